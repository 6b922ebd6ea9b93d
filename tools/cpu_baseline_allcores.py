#!/usr/bin/env python3
"""The oracle on every host core (CPU baseline plan (ii)): one utterance stream per process, same workload as
bench.py's cpu_baseline leg.  Context for the GPU number only."""
import os, sys, time
for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"): os.environ.setdefault(v, "1")   # one thread per process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
import numpy as np

def work(args):
    rank, n_utt, T = args
    from oracle import oracle_py as O
    from radae_amd.channel_tools import multipath_g, synth_features
    m = O.Model(); n_mf = T // 12
    sigma = float(O.lib().orc_sigma_from_EbNodB(3.0))
    rng = np.random.default_rng(100 + rank)
    t0 = time.perf_counter()
    for u in range(n_utt):
        feats = synth_features(1000 + rank * 64 + u, T)
        tx = O.Tx(m)
        sig = np.concatenate([tx.frame(feats[12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)])
        G = multipath_g("mpp", 8000, len(sig), 5000 + rank * 64 + u)
        n = len(sig)
        nz = ((rng.standard_normal(n + 1152) + 1j * rng.standard_normal(n + 1152)) / np.sqrt(2)).astype(np.complex64)
        r, fin = O.channel(sig, G, nz[:n], sigma, -11.0)
        e = O.channel_eoo(tx.eoo(), nz[n:], sigma, -11.0, 0.0, fin)
        full = np.concatenate([sigma * rng.standard_normal(8000), r, e, sigma * rng.standard_normal(1152)]).astype(np.complex64)
        O.run_rx_stream(m, full)
    return time.perf_counter() - t0

if __name__ == "__main__":
    from oracle import oracle_py as O
    O.build()
    T, n_utt = 1008, int(os.environ.get("UTT_PER_CORE", "8"))
    cores = int(os.environ.get("PROCS", len(os.sched_getaffinity(0))))
    t0 = time.perf_counter()
    with mp.Pool(cores) as pool:
        times = pool.map(work, [(r, n_utt, T) for r in range(cores)])
    el = time.perf_counter() - t0
    print(f"{cores} processes x {n_utt} utterances x {T} frames in {el:.1f} s wall: {cores * n_utt * T / el:.0f} frames/s on {cores} cores ({n_utt * T / np.mean(times):.0f} per core)")
