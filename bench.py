#!/usr/bin/env python3
"""bench.py -- vocoder-feature frames/s through encode -> OFDM mod -> MPP channel -> sync/demod/EQ -> decode.

Workload (BASELINE.json configs[2]): 256 synthetic utterances per GPU, 1008 feature
frames (84 modem frames, 10.08 s) each, model19_check3 weights, MPP Doppler-spread two-path channel,
AWGN at Eb/No = 3 dB, -11 Hz offset, 1 s of noise prepended, EOO frame + 1152 samples appended.
A "step" = that whole batch once, starting from reset encoder/receiver state, inputs (features, G)
already resident in HBM.  `--gpus N` > 1: one process per GPU -- started by this script itself under
torch.distributed.run, or by the caller (WORLD_SIZE must then equal --gpus) -- utterances sharded with no data-path
collective; the only collective is the RCCL broadcast of the weight blob.
Defaults: the receiver kernel with two streams per CU (k_rx_sync2) and three batches in flight.

Prints ONE JSON line on rank 0 (see the task contract): value = whole-job frames/s.
`--config 2` instead measures BASELINE.json configs[1] (one stream through the rade_core.h-level encoder / decoder,
latency-bound by construction) and prints its own line.

roofline block: `frac` prices the dominant kernel's work in its CHEAPEST known formulation (pilot search as one
|Dt| surface by FFT convolution, decoder, in-sync DSP -- the constants below) at the f32 peak, over the time that kernel is busy in
the TIMED configuration (launches of the batches in flight overlap); `alone` = one launch by itself, `whole_job` = every kernel's
work x frames/s.  The receiver kernel actually evaluates the search surface as split-binary16 GEMMs on the matrix cores (since round 5 in
two stages -- 16 polynomial moments, then their expansion to the 40 frequencies: 28 MFLOP f32-equivalent per call, 49 in rounds 3-4 --
instead of the FFT form's 5.1): pricing THAT would inflate the fraction several-fold for the search calls, so the FFT count stays.  The reference-formulation figure (98 MFLOP of GEMM per search call) is reported separately as
`equiv_ref_formulation` and is not a roofline.
"""
import argparse
import json
import os
import resource
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch

NMF = 960
F32_PEAK_TFLOPS = 157.3          # MI355X f32 matrix == f32 vector peak (MI355X_MICROARCH.md)
F16_PEAK_TFLOPS = 2500.0         # dense binary16 MFMA peak (MI355X_MICROARCH.md; the sparsity-inclusive headline figure is not used)
DTYPE = "f32 I/O, 2xf16 (22-bit) MFMA operands, f32 accumulate (refine: f64 MFMA)"
HBM_PEAK_GBS = 8000.0
# ---- executed-work model of k_rx_sync (SURVEY.md 8d figures) --------------------------------------------
PREWARM_SECONDS = 1.5                            # see main(): untimed, before the W warm-up steps (the clocks of an idle GPU take about a second of load to settle)
# SURVEY 8d's 866,560 cMAC per modem frame include the modulator's 24,000-cMAC IDFT (k_ofdm_mod) and the 96,960-cMAC band-pass filter (k_rx_bpf), neither in the receiver kernel
BPF_CMAC_PER_SAMPLE = 101                        # complex_bpf: 101 real taps on a complex sample; runs in k_rx_bpf ahead of the receiver kernel since round 4
RX_SYNC_CMAC = 866560 - 24000 - 960 * BPF_CMAC_PER_SAMPLE   # the receiver kernel's share: refine, check_pilots, demodulator DFT (745,600 cMAC)
REF_SYNC_CALL_FLOP = RX_SYNC_CMAC * 8.0          # in-sync DSP of the receiver kernel per modem frame as the reference formulates it (8 flop per cMAC): 5.96 MFLOP
# executed: refine() in sync runs as 8 moments x 16 timings x 2 frames x 160 samples (40,960 cMAC) + the polynomials (640 x ~40 flop)
# instead of 20 frequencies x 16 x 2 x 160 (102,400 cMAC): 0.47 MFLOP less per call
SYNC_CALL_FLOP = (RX_SYNC_CMAC - 102400 + 40960) * 8.0 + 640 * 40.0     # 5.50 MFLOP
ENC_STEP_FLOP = 2.0 * (96 * 64 + (64 + 224 + 384 + 544 + 704) * 192 + 2 * (128 + 288 + 448 + 608 + 768) * 96 + 864 * 80 + 5 * 64 * 192)   # CoreEncoder, one 40 ms step (4 feature frames): 1.87 MFLOP
DEC_MF_FLOP = 3 * 904064 * 2.0                   # CoreDecoder, 3 steps per decoded modem frame (runs inside k_rx_sync): 5.42 MFLOP = 0.452 MFLOP per feature frame
BPF_SAMPLE_FLOP = BPF_CMAC_PER_SAMPLE * 8.0      # band-pass pre-pass (k_rx_bpf), every received sample whatever the sync state: counted in whole_job, not in the receiver kernel
FFT_SURFACE_FLOP = 41 * 5.0 * 2048 * 11 + 40 * 2048 * 6.0   # one |Dt| surface by FFT convolution: 1 forward + 40 inverse 2048-point FFTs (5 N log2 N) + 40 spectral products: 5.11 MFLOP (the cheapest formulation; both receiver kernels now run the 49-MFLOP GEMM form on the matrix cores and are still priced at this figure)
REF_SEARCH_CALL_FLOP = 960 * 40 * 160 * 2 * 8.0  # the reference's formulation of detect_pilots (two surfaces as GEMMs): 98.3 MFLOP -- NOT executed here
ALGO_BYTES_PER_FRAME = 4128                      # whole path, BASELINE.md section 4
RX_ALGO_BYTES_PER_FRAME = 640 + 144              # the receiver kernel's share: IQ in + features out
RX_KERNEL_NAME = "k_rx_sync2"                     # the receiver kernel (rade_rx.hip): the PMC summary is looked up under this name
PROFILE_TAG = "r06"                              # profiles/<tag>_pmc_summary.json etc. (tools/collect_profiles.sh)


def executed_flop(search_calls, sync_calls, decoded_mf):
    return sync_calls * SYNC_CALL_FLOP + decoded_mf * DEC_MF_FLOP + search_calls * FFT_SURFACE_FLOP


def launch_plan(gpus, env, n_visible, argv, port=None, oversubscribe=None):
    """What `bench.py --gpus N` has to do before any GPU work (pure function: the CPU tests call it).
    Returns ("inprocess", None) -- this process is the job (N == 1) or one rank of it (started by torch.distributed.run: WORLD_SIZE
    must then equal --gpus) -- or ("spawn", argv_of_child): N > 1 asked for by a plain `python bench.py --gpus N`, so this process
    re-executes itself under `python -m torch.distributed.run` with one rank per GPU and relays the ranks' output.
    Raises SystemExit (non-zero) when fewer than N devices are visible or the rank count disagrees with --gpus."""
    if gpus < 1:
        raise SystemExit(f"bench.py: --gpus {gpus} is not a GPU count")
    if oversubscribe is not None:
        # developer mode (--oversubscribe-device D): the N ranks of an N-GPU job, all on device D -- a HOST-contention proxy (N processes x 3 engines share the
        # node's CPUs the way a real node's ranks would), not a scaling measurement; the line says so and reports n_gpus = 1
        if n_visible < oversubscribe + 1:
            raise SystemExit(f"bench.py: --oversubscribe-device {oversubscribe} but only {n_visible} HIP device(s) are visible")
        if "WORLD_SIZE" in env:
            if int(env["WORLD_SIZE"]) != gpus:
                raise SystemExit(f"bench.py: --gpus {gpus} but the launcher started {env['WORLD_SIZE']} rank(s)")
            return "inprocess", None
        if gpus == 1:
            return "inprocess", None
        if port is None:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        return "spawn", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but the launcher started {world} rank(s) (WORLD_SIZE): n_gpus in the line would be wrong")
        if n_visible < int(env.get("LOCAL_RANK", "0")) + 1:
            raise SystemExit(f"bench.py: rank with LOCAL_RANK={env.get('LOCAL_RANK', '0')} has no GPU ({n_visible} HIP device(s) visible)")
        return "inprocess", None
    if gpus == 1:
        if n_visible < 1:
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        return "inprocess", None
    if n_visible < gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} asked for but only {n_visible} HIP device(s) are visible -- refusing to report a {gpus}-GPU number from fewer GPUs")
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    return "spawn", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
                     "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def cpu_quota():
    """(logical CPUs visible to this process, cgroup CPU quota in cores or None)"""
    ncpu = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return ncpu, (None if q == "max" else float(q) / float(per))
    except Exception:
        return ncpu, None


def sync_peers(world, env):
    """$RADE_SYNC_PEERS for a rank of a `world`-rank job (None: leave the environment alone): the ranks of a node share its CPUs, so the library's
    wait policy must count every rank's engines -- the caller's value wins, else torchrun's LOCAL_WORLD_SIZE, else the world size (one node)."""
    if world <= 1:
        return None
    return env.get("RADE_SYNC_PEERS") or env.get("LOCAL_WORLD_SIZE") or str(world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--repeats", type=int, default=3, help="the timed K-step loop is run this many times (barrier-bracketed each); value = the median repeat")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=1008, help="10 ms feature frames per utterance (multiple of 12)")
    ap.add_argument("--config", type=int, default=3, choices=(2, 3), help="3: the headline batch workload; 2: single-stream core encoder/decoder latency")
    ap.add_argument("--pipeline", type=int, default=3, help="batches in flight: engines + HIP streams + host threads that take the steps in turn (1 = one batch at a time)")
    ap.add_argument("--two-pass-channel", action="store_true", help="rade_batch_tx + rade_batch_channel as two calls (k_chan_power + k_chan_apply) instead of rade_batch_tx_channel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--as-shard", type=str, default=None, metavar="R/N", help="developer mode: a single process takes the utterance shard rank R of an N-rank job would own (its results must equal that rank's)")
    ap.add_argument("--oversubscribe-device", type=int, default=None, metavar="D",
                    help="developer mode: run the --gpus N ranks of an N-GPU job all on HIP device D (gloo for the two collectives): what N processes x 3 engines cost the HOST; not a scaling measurement")
    args = ap.parse_args()
    if args.pipeline > 3:                # HIP multiplexes a process's streams onto 4 hardware queues per device by default; streams that share one serialise
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    mode, child = launch_plan(args.gpus, os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0, sys.argv[1:], oversubscribe=args.oversubscribe_device)
    if mode == "spawn":                  # `python bench.py --gpus N`: one rank per GPU under torch.distributed.run; rank 0 of the child prints the line
        import subprocess
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env["RADE_BENCH_LAUNCHER"] = "self (bench.py re-executed under torch.distributed.run)"
        raise SystemExit(subprocess.call(child, env=env))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, (world, args.gpus)
    launcher = os.environ.get("RADE_BENCH_LAUNCHER", "torch.distributed.run (caller)" if "WORLD_SIZE" in os.environ else "in-process, single rank")
    oversub = args.oversubscribe_device is not None
    if oversub:
        local = args.oversubscribe_device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = torch.device("cpu") if oversub else dev       # where the two collectives' tensors live (gloo when every rank sits on one device: RCCL wants a device per rank)
    if args.config == 2:
        print(json.dumps(config2(args.frames)))
        return
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if oversub: dist.init_process_group("gloo", rank=rank, world_size=world)
        else: dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from radae_amd.channel_tools import multipath_g, synth_features
    from radae_amd.engine import BatchEngine, DEFAULT_BLOB, sigma_from_EbNodB
    from radae_amd.parallel import broadcast_blob, gather_stats, shard_range

    B, T = args.streams, args.frames
    n_mf = T // 12
    n_sig, n_pre, n_post = n_mf * NMF, 8000, 1152

    # ---- weights: rank 0 reads the blob, every other rank receives it over RCCL/xGMI
    blob = broadcast_blob(DEFAULT_BLOB if rank == 0 else None, coll_dev, world)
    # `--pipeline` engines, each with its own state, HIP stream and host thread: the steps are dealt to them in turn, so the next batch's
    # encoder / channel kernels (and the head of its receiver launch) fill the CUs that the slowest streams of the previous batch's receiver
    # launch leave idle (a receiver launch lasts as long as its slowest stream; rade_batch_rx synchronises its stream, hence one thread each)
    depth = max(1, min(args.pipeline, args.steps))
    # one process per GPU: the ranks of a node share the container's CPUs, and the library's wait policy (spin while engines <= CPUs, else sleep on a
    # blocking event) only sees this process's engines -- tell it how many such processes there are (8 ranks x 3 engines spinning under a 16-core
    # quota would throttle each other)
    peers = sync_peers(world, os.environ)
    if peers is not None:
        os.environ["RADE_SYNC_PEERS"] = peers
    # developer switch (tools/ab_bench.py: `spec=,RADE_BENCH_SPEC_DEC=1`): the OPTIMISTIC BOUND of a speculative batched decoder -- the receiver
    # launch emits latents only (RADE_BATCH_BYPASS_DEC: no decoder, no UW accounting in the chain), the layer-wise batched decoder (rade_batch_decode) then decodes
    # every stream's rows; no UW evaluation, no replay of the streams whose UW count would have ended sync, no decoder resets: NOT the product path (results differ),
    # a ceiling on what that structure could reach
    spec_dec = bool(os.environ.get("RADE_BENCH_SPEC_DEC"))
    from radae_amd.engine import BYPASS_DEC
    engs = [BatchEngine(B, max_tx_mf=n_mf, device=local, blob_bytes=blob, flags=BYPASS_DEC if spec_dec else 0) for _ in range(depth)]
    eng = engs[0]
    lanes = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    out_bufs = [(torch.zeros((B, (n_pre + n_sig + 1152 + n_post) // 800 + 1, 240 if spec_dec else 432), dtype=torch.float32, device=dev), torch.zeros((B, 180), dtype=torch.float32, device=dev)) for _ in range(depth)]

    # ---- synthetic inputs, resident in HBM before the clock starts.  Utterance u uses seeds 1000 + u / 5000 + u; rank r owns the
    # contiguous shard [r B, r B + B) of the B x world utterances
    u0, u1 = shard_range(B * world, rank, world)
    if args.as_shard:
        assert world == 1
        sr, sn = (int(v) for v in args.as_shard.split("/"))
        u0, u1 = shard_range(B * sn, sr, sn)
    assert u1 - u0 == B
    feats_np = np.stack([synth_features(1000 + u, T) for u in range(u0, u1)])
    feats = torch.tensor(feats_np, device=dev)
    G = torch.empty((B, n_sig, 2), dtype=torch.complex64, device=dev)
    for b in range(B):
        G[b] = torch.from_numpy(multipath_g("mpp", 8000, n_sig, 5000 + u0 + b)).to(dev)
    sigma = sigma_from_EbNodB(3.0)

    def step(seed, e=None):
        e = e or eng
        e.reset()
        if args.two_pass_channel:
            rx = e.channel(e.tx(feats), sigma, -11.0, n_pre=n_pre, n_post=n_post, with_eoo=True, G=G, seed=seed)
        else:           # transmit + channel in one pass (rade_batch_tx_channel: the modulator applies the two-path model, no second pass over tx and G)
            rx = e.tx_channel(feats, sigma, -11.0, n_pre=n_pre, n_post=n_post, with_eoo=True, G=G, seed=seed)
        # the receiver's outputs go to caller-owned buffers, one set per engine as in hosts/rade_multi_bench.c (rows beyond status.n_valid keep what they
        # held: the C ABI never promised zeroes); without them the Python binding allocates and zero-fills 37 MB per step, two more launches in the lane
        i = engs.index(e)
        if spec_dec:
            z, st_, eo_ = e.rx(rx, features_out=out_bufs[i][0], eoo_out=out_bufs[i][1])
            rows = 3 * max(s.n_valid for s in st_)
            fo_ = e.decode(z.view(B, -1, 80)[:, :rows].contiguous(), 84) if rows else z
            return fo_, st_, eo_, rx
        return e.rx(rx, features_out=out_bufs[i][0], eoo_out=out_bufs[i][1]) + (rx,)

    def run_steps(n, seed0):
        """steps seed0 .. seed0 + n - 1, dealt round-robin to the lanes; returns the results of the last step"""
        import threading
        last = [None] * depth
        def lane(i):
            torch.cuda.set_device(local)
            with torch.cuda.stream(lanes[i]):
                for k in range(i, n, depth):
                    last[i] = step(seed0 + k, engs[i])
        ths = [threading.Thread(target=lane, args=(i,)) for i in range(min(depth, n))]
        [t.start() for t in ths]; [t.join() for t in ths]
        return last[(n - 1) % depth]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    # untimed set-up on top of the W warm-up steps the caller asks for: a freshly started GPU box runs its first few dozen launches at
    # lower clocks / with cold code and page tables (measured: 45.6 M frames/s for a whole 100-step run that started cold, 49.1 M
    # for the same command right after another run), so the engines are exercised for PREWARM_SECONDS before anything is counted (24 steps were not enough)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < PREWARM_SECONDS:
        run_steps(8, 200)
    if args.warmup:
        run_steps(args.warmup, 100)
    # the timed region: EXACTLY K steps between barrier + synchronize on both sides, max over ranks; run `--repeats` times (same seeds, so
    # the same work) and the median repeat is the one reported -- every repeat is in the line (value_repeats)
    reps = []
    import gc
    for _ in range(max(1, args.repeats)):
        # (the interpreter's cyclic garbage collector is collected before and off inside the timed region, as timeit does it: a generation-2 pause with
        # torch imported is 35-40 ms of the MEASURING script holding the GIL -- no lane thread launches anything meanwhile; bench.py --config 2 met one per run)
        gc.collect(); gc_was = gc.isenabled(); gc.disable()
        barrier()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        fo, st, _, rx_last = run_steps(args.steps, 1)
        barrier()
        dt_local = time.perf_counter() - t0
        if gc_was: gc.enable()
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
        fo = fo.clone()                  # (outside the clock) the engine's own output buffer is reused by the measurement legs below; the parity sample wants the timed step's
        dt = dt_local
        per_rank_ms = [1e3 * dt_local / args.steps]
        if world > 1:
            import torch.distributed as dist
            tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
            allms = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
            dist.all_gather(allms, torch.tensor([1e3 * dt_local / args.steps], dtype=torch.float64, device=coll_dev))
            per_rank_ms = [float(x.item()) for x in allms]
        reps.append((dt, per_rank_ms, cpu_s))
    dt, per_rank_ms, cpu_s = sorted(reps, key=lambda r: r[0])[(len(reps) - 1) // 2]      # median (lower middle for an even count)
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        ranks_seen = dist.get_world_size()

    total_frames = B * T * args.steps * world
    value = total_frames / dt
    # job-wide statistics of the last step (the only other collective besides the blob broadcast: one small all-reduce)
    nv = np.array([s.n_valid for s in st]); ncalls = np.array([s.n_calls for s in st]); neoo = np.array([s.has_eoo for s in st])
    sync_calls = int((nv + neoo).sum()); search_calls = int(ncalls.sum()) - sync_calls
    job = gather_stats(np.array([B * T, 12.0 * nv.sum(), ncalls.sum(), sync_calls, search_calls, neoo.sum()], dtype=np.float64), coll_dev, world)
    # what the last timed step decoded on this rank, as a digest (outside the clock): rank 0 of an N-rank job owns the same utterances as a single-rank job,
    # so its digest must equal the single-rank run's (tools/oversub8.sh checks it)
    import hashlib
    dig = hashlib.sha256()
    fo_host = fo.cpu().numpy()
    for b in range(B):
        dig.update(np.ascontiguousarray(fo_host[b, :st[b].n_valid]).tobytes())
    host_cpu = None; all_digests = [dig.hexdigest()]
    if world > 1:
        import torch.distributed as dist
        mine = torch.tensor([cpu_s / dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        host_cpu = float(mine.item())
        dg = [torch.zeros(32, dtype=torch.uint8, device=coll_dev) for _ in range(world)]
        dist.all_gather(dg, torch.tensor(list(dig.digest()), dtype=torch.uint8, device=coll_dev))
        all_digests = [bytes(d.cpu().tolist()).hex() for d in dg]

    out = {
        "metric": "vocoder-feature frames/sec (enc+chan+dec), model19", "value": value, "unit": "frames/s", "n_gpus": 1 if oversub else world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "model19_check3 streaming radae_txe -> OFDM + MPP multipath/AWGN 3 dB/-11 Hz -> radae_rxe (configs[2])",
                   "streams_per_gpu": B, "frames_per_stream": T, "global_streams": B * world, "batches_in_flight": depth, **({"developer_variant": "RADE_BENCH_SPEC_DEC: latents-only receiver + batched decoder, no UW / replay (optimistic bound, not the product path)"} if spec_dec else {}), "parallelism": f"utterance-sharded x{world}, RCCL weight broadcast only",
                   "arithmetic": "f32 DSP, f64 refine, matrix products on split-binary16 (2 x 11 bit) MFMA with f32 accumulation"},
        "pipelining": f"{depth} batch(es) in flight per GPU (engines with their own state on their own HIP stream / host thread, steps dealt in turn); ms_per_step = wall time / steps; "
                      "roofline.sum_kernel_ms_per_step is one batch alone",
        "value_counts": "offered feature frames: every transmitted frame's samples pass through the receiver, decoded or not",
        "decoded_frames_per_s": value * job[1] / job[0],
        "timed_region_s": dt, "value_repeats": [B * T * args.steps * world / r[0] for r in reps], "timed_region_s_repeats": [r[0] for r in reps],
        "launcher": launcher, "rccl_ranks": ranks_seen, "last_step_features_sha256_rank0": all_digests[0], "last_step_features_sha256_per_rank": all_digests, "utterances": [u0, u1],
        **({"oversubscribed": {"ranks": world, "device": local, "collectives": "gloo", "cpu_cores_busy_all_ranks": host_cpu,
                               "note": "developer mode: the ranks of a --gpus N job all on ONE device: a host-contention proxy (N processes x engines share the CPUs as a node's ranks would), NOT a scaling measurement; value = aggregate of the ranks sharing that one GPU"}} if oversub else {}),
        "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
        "job_last_step": {"offered_frames": int(job[0]), "decoded_frames": int(job[1]), "rx_calls": int(job[2]), "sync_calls": int(job[3]),
                          "search_calls": int(job[4]), "eoo_detected_streams": int(job[5])},
    }
    if rank == 0:
        # host side of rank 0: CPU time its threads burnt per step (a spinning wait counts in full), the CPUs it may use, and how its engines waited
        # (rade_batch_rx spins while engines <= CPUs, else sleeps: include/rade_batch.h; $RADE_SYNC overrides)
        ncpu, quota = cpu_quota()
        sc = [e.sync_counts() for e in engs]
        out["host"] = {"cpu_s_per_step": cpu_s / args.steps, "cpu_cores_busy": cpu_s / dt, "logical_cpus": ncpu, "cgroup_cpu_quota": quota,
                       "engines_in_process": depth, "rx_waits_blocking": sum(a for a, _ in sc), "rx_waits_spinning": sum(b for _, b in sc),
                       "RADE_SYNC": os.environ.get("RADE_SYNC", "auto"), "RADE_SYNC_PEERS": int(os.environ.get("RADE_SYNC_PEERS", "1"))}
        out["decoded_modem_frames_per_stream"] = {"min": int(nv.min()), "mean": float(nv.mean()), "max": int(nv.max())}

    if rank == 0 and not args.no_roofline:
        out["roofline"] = roofline_leg(eng, step, args.steps, B, T, value, world)
        out["roofline"]["rx_kernel"] = "k_rx_sync2 (256 threads / at most 80 KB of LDS per stream: two streams per CU)"
        if depth > 1 and out["roofline"].get("kernel") == "rx_sync":
            pipelined_roofline(out["roofline"], engs, run_steps, min(args.steps, 24), dev)
        if out["roofline"].get("kernel") == "rx_sync":
            two_per_cu_leg(out["roofline"], B, T, n_mf, local, blob, feats, G, sigma, n_pre, n_post)
    if rank == 0 and not args.no_parity:
        out["parity_sample"] = parity_leg(feats_np, fo, st, rx_last, blob, local, B)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # configs[1] (single stream through rade_core_encoder / rade_core_decoder) beside the headline workload, so the driver's record carries it
        # too: `python bench.py --config 2` prints the full line.  Ahead of the CPU baselines: the all-cores leg runs the container's CPU quota dry,
        # and a call of this latency measurement that started in the same scheduler period was then throttled for 35-60 ms (`decoder_call_max` of two
        # profile runs) -- a third of the loop's time
        c2 = config2(T)
        out["config2"] = {"workload": c2["config"]["workload"], "ms_per_step": c2["ms_per_step"], "frames_per_s": c2["value"], "latency_ms": c2["latency_ms"],
                          "cpu_frames_per_s": c2["cpu_baseline"]["value"], "cpu_sample": c2["cpu_baseline"]["sample"], "speedup_vs_one_core": c2["value"] / c2["cpu_baseline"]["value"],
                          "parity": c2["parity"]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(feats_np, T)
        out["cpu_baseline_allcores"] = cpu_baseline_allcores(T)

    if rank == 0:
        print(json.dumps(out))
    for e in engs:
        e.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                   # rank 0 may still be in its extra legs: leave the group together
        dist.destroy_process_group()


def roofline_leg(eng, step, steps, B, T, value, world):
    """Per-kernel-class HIP-event timing of the same K steps again (same noise seeds: the receiver kernel's duration is
    the slowest stream's and moves by +-5 % with the seed).  Events are recorded on the launch stream and read back
    afterwards, so these steps run back to back like the timed ones."""
    steps = min(steps, 25)               # the event pool of the engine is drained every 256 launches; 25 steps is plenty for an average
    eng.profile(True)
    search = sync = dec_mf = 0
    for k in range(steps):
        _, st, _, _ = step(1 + k)
        s_sync = sum(s.n_valid + s.has_eoo for s in st)
        sync += s_sync; search += sum(s.n_calls for s in st) - s_sync; dec_mf += sum(s.n_valid for s in st)
    torch.cuda.synchronize()
    eng.profile(False)
    prof = eng.profile_get()
    dom = max(prof, key=lambda k: prof[k]["ms"])
    p = prof[dom]
    launches = max(p["launches"], 1)
    counts = {"search_calls": search / launches, "sync_calls": sync / launches, "decoded_modem_frames": dec_mf / launches, "offered_frames": B * T}
    r = {"kernel": dom, "avg_launch_ms": p["ms"] / launches, "launches_per_step": p["launches"] / steps, "steps_profiled": steps,
         "per_class_ms_per_step": {k: round(v["ms"] / steps, 4) for k, v in prof.items()},
         "sum_kernel_ms_per_step": round(sum(v["ms"] for v in prof.values()) / steps, 4),
         "per_class_launches_per_step": {k: v["launches"] / steps for k, v in prof.items()},
         "hbm_frac_whole_job": value / world * ALGO_BYTES_PER_FRAME / (HBM_PEAK_GBS * 1e9)}
    if dom == "rx_sync":
        fl = executed_flop(counts["search_calls"], counts["sync_calls"], counts["decoded_modem_frames"])
        achieved = fl / (r["avg_launch_ms"] * 1e-3) / 1e12
        algo_bytes = RX_ALGO_BYTES_PER_FRAME * B * T
        n_rx_samples = 8000 + (T // 12) * NMF + 1152 + 1152                                       # per stream: noise prefix, signal, EOO frame, tail
        step_flop = fl + ENC_STEP_FLOP * B * T / 4 + 8.0 * B * (T // 12) * 5 * 30 * 160 + BPF_SAMPLE_FLOP * B * n_rx_samples   # receiver + encoder + modulator IDFT + band-pass pre-pass, as executed
        r.update({"achieved": achieved, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F32_PEAK_TFLOPS,
                  "whole_job": {"executed_flop_per_step": step_flop, "executed_flop_per_frame": step_flop / (B * T), "achieved": value / world * step_flop / (B * T) / 1e12,
                                "frac": value / world * step_flop / (B * T) / 1e12 / F32_PEAK_TFLOPS,
                                "note": "every kernel's executed work (receiver as above + CoreEncoder GEMMs / recurrences + modulator) x the timed frames/s, against the same f32 peak: the figure that does not depend on how the kernels overlap"},
                  "per_launch_counts": counts, "executed_flop_per_launch": fl, "algorithmic_bytes_per_launch": algo_bytes,
                  "flop_model": {"sync_call": SYNC_CALL_FLOP, "decoded_modem_frame": DEC_MF_FLOP, "search_call": FFT_SURFACE_FLOP, "bpf_sample_whole_job_only": BPF_SAMPLE_FLOP,
                                 "note": "executed work of the receiver kernel: in-sync DSP 684,160 cMAC x 8 (refine by moments: 40,960 cMAC instead of the reference formulation's 102,400; the 96,960-cMAC band-pass filter runs in k_rx_bpf and is priced in whole_job only) + polynomials per synchronised call, decoder 3 x 904,064 MAC x 2 per decoded modem frame, "
                                         "search call = one |Dt| surface PRICED as FFT convolution (41 x 5 N log2 N + 40 x 6 N, N = 2048: the cheapest formulation; the kernels evaluate it as a split-binary16 GEMM on the matrix cores, 49 MFLOP f32-equivalent, which is not counted) ; priced at the f32 peak"},
                  "hbm_frac_kernel": algo_bytes / (r["avg_launch_ms"] * 1e-3) / (HBM_PEAK_GBS * 1e9),
                  "equiv_ref_formulation": {"tflops": (counts["sync_calls"] * REF_SYNC_CALL_FLOP + counts["decoded_modem_frames"] * DEC_MF_FLOP + counts["search_calls"] * REF_SEARCH_CALL_FLOP)
                                            / (r["avg_launch_ms"] * 1e-3) / 1e12,
                                            "note": "detect_pilots priced as the reference's two 960x40x160 complex GEMMs (98.3 MFLOP per call); work NOT executed in that form -- not a roofline"}})
    else:
        achieved = p["flops"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
        r.update({"achieved": achieved, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F32_PEAK_TFLOPS})
    # counters of the same command, collected in separate rocprofv3 --pmc passes (tools/collect_profiles.sh -> profiles/).
    # (achieved, peak, frac) keep the contract's pricing -- executed work in its cheapest formulation against the f32 matrix peak ("priced_against") -- and
    # describe ONE launch with the chip to itself, the duration `rocprofv3 --kernel-trace --stats` reproduces (profiles/<tag>_bench_kernel_stats.csv).
    # "bound" says what limits the kernel: "mfma" / "hbm" when a pipe is, "latency" when neither is (matrix pipe < 25 % busy by SQ_BUSY counters AND
    # algorithmic bytes < 10 % of HBM): a per-stream serial chain parked on s_waitcnt / s_barrier, then VALU issue ("limiter", SQ wave-cycle shares).
    # f16_pipe prices the matrix instructions the kernel actually ISSUES (SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 flop, split-binary16 products) against the
    # dense binary16 peak: the distance to the pipe the products run on.
    r["priced_against"] = "mfma (f32 matrix peak; SURVEY 8(d))"
    r["bound"] = "mfma"
    r["limiter"] = "latency (s_waitcnt/s_barrier) + valu-issue"
    r["traffic"] = None
    try:
        tag = next(t for t in (PROFILE_TAG, "r05", "r04", "r03") if os.path.exists(os.path.join(REPO, "profiles", f"{t}_pmc_summary.json")))
        pm = json.load(open(os.path.join(REPO, "profiles", f"{tag}_pmc_summary.json")))
        k = pm["kernels"][{"rx_sync": RX_KERNEL_NAME}.get(dom, dom)]
        raw = k["fetch_bytes_per_dispatch"] + k["write_bytes_per_dispatch"]
        r["traffic"] = raw
        r["traffic_fetch_wide_corrected"] = 2.0 * k["fetch_bytes_per_dispatch"] + k["write_bytes_per_dispatch"]   # gfx950: FETCH_SIZE halves 16-byte-per-lane loads (upper bound: not every load is that wide)
        if dom == "rx_sync":
            r["traffic_ratio"] = raw / (RX_ALGO_BYTES_PER_FRAME * B * T)
            r["traffic_ratio_wide_corrected"] = r["traffic_fetch_wide_corrected"] / (RX_ALGO_BYTES_PER_FRAME * B * T)
        tb = pm.get("rx_sync_traffic_breakdown_bytes_per_launch")
        if tb and dom == "rx_sync":
            r["traffic_ratio_incl_search_state"] = tb["ratio_raw_over_io_plus_search_state"]      # |Dt| surfaces kept between search calls counted as algorithmic state
            r["traffic_breakdown"] = f"profiles/{tag}_pmc_summary.json:rx_sync_traffic_breakdown_bytes_per_launch"
        if "l2_hit_rate" in k:
            r["l2_hit_rate_pmc"] = k["l2_hit_rate"]
        r["mfma_busy_pct_pmc"] = k["mfma_busy_pct"]
        if dom == "rx_sync" and "mfma_mops_f16_per_dispatch" in k:
            f16 = 512.0 * k["mfma_mops_f16_per_dispatch"]
            ach16 = f16 / (r["avg_launch_ms"] * 1e-3) / 1e12
            r["f16_pipe"] = {"executed_f16_flop_per_launch": f16, "achieved": ach16, "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach16 / F16_PEAK_TFLOPS,
                             "f64_flop_per_launch": 512.0 * k.get("mfma_mops_f64_per_dispatch", 0.0), "f32_flop_per_launch": 512.0 * k.get("mfma_mops_f32_per_dispatch", 0.0),
                             "note": "issued matrix work of one launch (PMC pass of the profiled command: SQ_INSTS_VALU_MFMA_MOPS_F16 x 512) over this run's launch duration, against the dense binary16 peak"}
        if k["mfma_busy_pct"] < 25.0 and r.get("hbm_frac_kernel", 1.0) < 0.1:
            r["bound"] = "latency"
        r["counters_from"] = f"profiles/{tag}_pmc_summary.json (commit {pm.get('commit', '?')})"
        sq = pm.get("sq_breakdown", {}).get(RX_KERNEL_NAME)
        if sq:
            r["sq_wave_cycle_shares"] = sq
            r["limiter"] = sq.get("bound", r["limiter"])
    except Exception:
        pass
    return r


def pipelined_roofline(r, engs, run_steps, n, dev):
    """The dominant kernel as it runs in the TIMED configuration: with several batches in flight the receiver launches of different engines
    overlap on the device (k_rx_sync2: two workgroups per CU), so one launch's duration says little about the chip.  Every receiver launch
    of `n` more pipelined steps (same seeds as the leg above, hence the same per-launch work) is put on one time axis with HIP events
    (rade_batch_profile_intervals); achieved = executed FLOP of all of them / the time during which at least one was running."""
    ref = torch.cuda.Event(enable_timing=True); ref.record(); torch.cuda.synchronize()
    for e in engs:
        e.profile_ref(ref.cuda_event); e.profile(True)
    run_steps(n, 1)
    torch.cuda.synchronize()
    iv = []
    for e in engs:
        e.profile(False)
        t0, t1 = e.profile_intervals("rx_sync")
        iv += list(zip(t0.tolist(), t1.tolist()))
        e.profile_ref(0)
    iv.sort()
    union, cur0, cur1 = 0.0, None, None
    for a, b in iv:
        if cur1 is None or a > cur1:
            if cur1 is not None: union += cur1 - cur0
            cur0, cur1 = a, b
        else:
            cur1 = max(cur1, b)
    if cur1 is not None: union += cur1 - cur0
    total = sum(b - a for a, b in iv)
    fl = r["executed_flop_per_launch"] * len(iv)
    ach = fl / (union * 1e-3) / 1e12
    r["alone"] = {"avg_launch_ms": r["avg_launch_ms"], "achieved": r["achieved"], "frac": r["frac"], "note": "one launch with the chip to itself: the headline (achieved, frac, avg_launch_ms) of this block"}
    r["pipelined"] = {"launches": len(iv), "avg_launch_ms_overlapped": total / max(len(iv), 1), "busy_union_ms_per_launch": union / max(len(iv), 1),
                      "mean_concurrency": total / union if union else 0.0, "achieved": ach, "frac": ach / F32_PEAK_TFLOPS,
                      "note": "receiver launches of all engines in flight on one time axis (HIP events, rade_batch_profile_intervals); achieved = launches x executed FLOP per launch / time with at least one receiver launch running (the timed configuration; a sub-field since round 6: the headline is `alone`, which rocprofv3's per-kernel average reproduces)"}


def two_per_cu_leg(r, B, T, n_mf, local, blob, feats, G, sigma, n_pre, n_post, steps=6):
    """The receiver kernel at the occupancy it is built for, in ONE launch: k_rx_sync2 is sized for two workgroups (streams) per CU, and a
    launch of B = 256 streams alone puts one on each of the 256 CUs (`alone`).  Here one engine carries 2 B streams (the same utterances
    twice; the channel's noise is keyed by stream index, so the two copies differ), so a single launch fills both slots of every CU with no other
    kernel on the chip: executed FLOP of that launch / its HIP-event duration."""
    from radae_amd.engine import BatchEngine
    e2 = BatchEngine(2 * B, max_tx_mf=n_mf, device=local, blob_bytes=blob)
    f2 = torch.cat([feats, feats]); G2 = torch.cat([G, G])

    def step2(seed):
        e2.reset()
        return e2.rx(e2.tx_channel(f2, sigma, -11.0, n_pre=n_pre, n_post=n_post, with_eoo=True, G=G2, seed=seed))
    step2(1); step2(2)
    torch.cuda.synchronize()
    e2.profile(True)
    search = sync = dec_mf = 0
    for k in range(steps):
        _, st, _ = step2(1 + k)
        s_sync = sum(s.n_valid + s.has_eoo for s in st)
        sync += s_sync; search += sum(s.n_calls for s in st) - s_sync; dec_mf += sum(s.n_valid for s in st)
    torch.cuda.synchronize()
    e2.profile(False)
    p = e2.profile_get()["rx_sync"]
    e2.close()
    ms = p["ms"] / max(p["launches"], 1)
    fl = executed_flop(search / steps, sync / steps, dec_mf / steps)
    ach = fl / (ms * 1e-3) / 1e12
    r["two_per_cu"] = {"streams_per_launch": 2 * B, "avg_launch_ms": ms, "executed_flop_per_launch": fl, "achieved": ach, "frac": ach / F32_PEAK_TFLOPS,
                       "ms_per_256_streams": ms / 2,
                       "note": "one launch of 2 x 256 streams alone on the chip: both stream slots of every CU filled by a single launch (the kernel's design occupancy), "
                               "same flop model; `alone` is the same kernel with one slot of every CU empty"}


def parity_leg(feats_np, fo, st, rx, blob, local, B):
    """The oracle on the timed workload itself: the received samples of a few streams of the LAST timed step (device Philox noise
    included) are copied back and run through the CPU oracle; the same samples are replayed through a small traced engine, which
    must reproduce the timed engine's features bit for bit and the oracle's per-call discrete outputs exactly."""
    from oracle import oracle_py as O
    from radae_amd.engine import BatchEngine
    from radae_amd.loss import find_loss
    O.build()
    m = O.Model()
    idx = sorted({0, B // 3, (2 * B) // 3, B - 1})
    rx_host = rx[idx].cpu().numpy()
    e2 = BatchEngine(len(idx), max_tx_mf=1, device=local, blob_bytes=blob, rx_trace_calls=128)
    fo2, st2, _ = e2.rx(rx[idx].contiguous())
    torch.cuda.synchronize()
    keys = ["state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]
    res = {"streams": idx, "discrete_equal": True, "timed_equals_replay_bitwise": True, "feat_rms_max": 0.0, "loss_gpu": [], "loss_oracle": [], "loss_delta_max": 0.0,
           "refine_near_ties": 0, "fmax_bit_equal": True, "calls_compared": 0, "decoded_modem_frames": []}
    for j, b in enumerate(idx):
        d = O.run_rx_stream(m, rx_host[j])
        t = e2.rx_trace(j)
        nvb = st[b].n_valid
        f_timed = fo[b, :nvb].cpu().numpy(); f_rep = fo2[j, :st2[j].n_valid].cpu().numpy()
        if st2[j].n_valid != nvb or not np.array_equal(f_timed, f_rep):
            res["timed_equals_replay_bitwise"] = False
        ok = all(np.array_equal(t[k], d[k]) for k in keys) and nvb == len(d["features_out"])
        res["calls_compared"] += int(len(d["ret"]))
        res["decoded_modem_frames"].append(int(nvb))
        if not ok:
            res["discrete_equal"] = False
            continue
        if not np.array_equal(t["fmax"], d["fmax"]):
            res["fmax_bit_equal"] = False          # the frequency estimate of every call is the same double on both sides since round 4
            res["refine_near_ties"] += 1           # (rounds 2-3 counted fmax moved by a grid step here and tolerated it: it was a device bug)
        if nvb:
            res["feat_rms_max"] = max(res["feat_rms_max"], float(np.sqrt(np.mean((f_timed - d["features_out"]) ** 2))))
            lg, _ = find_loss(feats_np[b], f_timed.reshape(-1, 36)); lo, _ = find_loss(feats_np[b], d["features_out"].reshape(-1, 36))
            res["loss_gpu"].append(float(lg)); res["loss_oracle"].append(float(lo)); res["loss_delta_max"] = max(res["loss_delta_max"], abs(float(lg) - float(lo)))
    e2.close()
    return res


def _oracle_inputs(feats_np, T, n, seed0=11):
    """Fixture generation for the CPU legs -- OUTSIDE their timed regions (the GPU's timed region has G resident too)."""
    from radae_amd.channel_tools import multipath_g
    rng = np.random.default_rng(seed0)
    n_sig = (T // 12) * NMF
    items = []
    for u in range(n):
        G = multipath_g("mpp", 8000, n_sig, 5000 + u)
        nz = ((rng.standard_normal(n_sig + 1152) + 1j * rng.standard_normal(n_sig + 1152)) / np.sqrt(2)).astype(np.complex64)
        items.append((feats_np[u % len(feats_np)], G, nz, rng.standard_normal(8000).astype(np.float32), rng.standard_normal(1152).astype(np.float32)))
    return items


def _oracle_utterance(O, m, sigma, item, T):
    f, G, nz, pre, post = item
    n_mf = T // 12
    tx = O.Tx(m)
    sig = np.concatenate([tx.frame(f[12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)])
    n = len(sig)
    r, fin = O.channel(sig, G, nz[:n], sigma, -11.0)
    e = O.channel_eoo(tx.eoo(), nz[n:], sigma, -11.0, 0.0, fin)
    full = np.concatenate([sigma * pre, r, e, sigma * post]).astype(np.complex64)
    O.run_rx_stream(m, full)


def cpu_baseline(feats_np, T, budget_s=12.0, n_max=48):
    """The oracle (plain-C restatement, oracle/) timed on one host core over a bounded sample of the same workload;
    inputs (features, Doppler samples, noise) are generated before the clock starts.
    kind 'port': the reference's own C core cannot be built here (needs xiph/opus)."""
    from oracle import oracle_py as O
    O.build()
    m = O.Model()
    sigma = float(O.lib().orc_sigma_from_EbNodB(3.0))
    items = _oracle_inputs(feats_np, T, n_max)
    done, t0 = 0, time.perf_counter()
    for it in items:
        _oracle_utterance(O, m, sigma, it, T)
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    el = time.perf_counter() - t0
    return {"value": done * T / el, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{done} utterance(s) x {T} frames through oracle enc+mod+MPP channel+rx+dec in {el:.1f} s, 1 thread, inputs pre-generated"}


def _allcores_worker(args):
    rank, n_utt, T = args
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    sys.path.insert(0, REPO)
    from oracle import oracle_py as O
    from radae_amd.channel_tools import synth_features
    m = O.Model()
    sigma = float(O.lib().orc_sigma_from_EbNodB(3.0))
    feats = np.stack([synth_features(1000 + rank * 8 + u, T) for u in range(n_utt)])
    items = _oracle_inputs(feats, T, n_utt, seed0=100 + rank)
    t0 = time.perf_counter()
    for it in items:
        _oracle_utterance(O, m, sigma, it, T)
    return t0, time.perf_counter()


def cpu_baseline_allcores(T, n_utt=6):
    """The same oracle, one utterance stream per process, on the host's cores (SURVEY.md 8d plan (ii)).  Context only."""
    import multiprocessing as mp
    ncpu = len(os.sched_getaffinity(0))
    best = None
    for procs in sorted({min(16, ncpu), min(64, ncpu)}):
        with mp.get_context("spawn").Pool(procs) as pool:
            spans = pool.map(_allcores_worker, [(r, n_utt, T) for r in range(procs)])
        el = max(s[1] for s in spans) - min(s[0] for s in spans)      # first start to last finish of the oracle loops (process start-up excluded)
        rate = procs * n_utt * T / el
        if best is None or rate > best["value"]:
            quota = cpu_quota()[1]
            best = {"value": rate, "unit": "frames/s", "cores": procs, "kind": "port",
                    "sample": f"{procs} processes x {n_utt} utterances x {T} frames in {el:.1f} s ({ncpu} logical CPUs visible, cgroup CPU quota "
                              + (f"{quota:.1f} cores -- the quota, not the oracle, caps this figure" if quota else "none") + ")"}
    return best


def config2(T):
    """BASELINE.json configs[1]: one stream through the rade_core.h-level encoder and decoder (include/rade_core.h:
    rade_core_encoder / rade_core_decoder, one 40 ms step per call, host buffers in and out -- the PCIe hop is inside the timed calls),
    with the oracle on one host core beside it.  One step = one launch of k_core_step (rade_core_step.hip): the whole layer stack for
    one stream in one workgroup; latency-bound by construction."""
    from radae_amd import core
    from radae_amd.channel_tools import synth_features
    from radae_amd.engine import DEFAULT_BLOB
    n_steps = T // 4
    f = synth_features(1000, T)
    rows = np.concatenate([f[:, :20], -np.ones((T, 1), np.float32)], axis=1).reshape(n_steps, 84).astype(np.float32)
    enc = core.CoreEncoder(DEFAULT_BLOB); dec = core.CoreDecoder(DEFAULT_BLOB)
    z = np.zeros((n_steps, 80), np.float32); fh = np.zeros((n_steps, 84), np.float32)
    for i in range(min(32, n_steps)):      # warm-up of both directions (the first call of each opens its device state: blob parse + upload)
        dec.step(enc.step(rows[i]))
    enc.reset(); dec.reset()
    # encoder and decoder alternate as in a live link (tx and rx side of a radio run concurrently), every call timed on its own
    te = np.zeros(n_steps); td = np.zeros(n_steps)
    # the interpreter's cyclic garbage collector is off inside the timed loop, as timeit does it: with torch imported a generation-2 collection is a
    # 35-40 ms pause of the MEASURING script (seen once per 252 steps in two profile runs: encoder_call_max / decoder_call_max), a third of the loop's time
    import gc
    gc_was = gc.isenabled(); gc.disable()
    t0 = time.perf_counter()
    for i in range(n_steps):
        a = time.perf_counter(); z[i] = enc.step(rows[i]); b = time.perf_counter(); fh[i] = dec.step(z[i]); c = time.perf_counter()
        te[i] = b - a; td[i] = c - b
    t2 = time.perf_counter()
    if gc_was: gc.enable()
    t1 = t0 + te.sum()
    from oracle import oracle_py as O
    O.build()
    m = O.Model(); oe = O.Encoder(m); od = O.Decoder(m)
    zo = np.zeros_like(z); fo = np.zeros_like(fh)
    c0 = time.perf_counter()
    for i in range(n_steps):
        zo[i] = oe.step(rows[i])
    c1 = time.perf_counter()
    for i in range(n_steps):
        fo[i] = od.step(zo[i])
    c2 = time.perf_counter()
    out = {"metric": "vocoder-feature frames/sec (core enc + dec, single stream), model19_check3", "value": T / (t2 - t0), "unit": "frames/s", "n_gpus": 1,
           "steps": n_steps, "warmup": min(32, n_steps), "ms_per_step": 1e3 * (t2 - t0) / n_steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": DTYPE, "data": "synthetic",
           "config": {"workload": "model19_check3 rade_core_encoder + rade_core_decoder, single stream, one 40 ms step per call (configs[1])", "frames": T},
           "latency_ms": {"encoder_call": 1e3 * float(te.mean()), "decoder_call": 1e3 * float(td.mean()), "encoder_call_median": 1e3 * float(np.median(te)), "decoder_call_median": 1e3 * float(np.median(td)),
                          "encoder_call_max": 1e3 * float(te.max()), "decoder_call_max": 1e3 * float(td.max())},
           "parity": {"z_rms": float(np.sqrt(np.mean((z - zo) ** 2))), "features_rms": float(np.sqrt(np.mean((fh - fo) ** 2)))},
           "cpu_baseline": {"value": T / (c2 - c0), "unit": "frames/s", "cores": 1, "kind": "port",
                            "sample": f"{n_steps} encoder + decoder steps of the oracle, 1 thread ({1e3 * (c1 - c0) / n_steps:.3f} + {1e3 * (c2 - c1) / n_steps:.3f} ms per step)"}}
    enc.close(); dec.close()
    return out


if __name__ == "__main__":
    main()
