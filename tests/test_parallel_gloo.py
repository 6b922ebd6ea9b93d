"""N > 1 path on CPU: world_size-2 gloo processes exercise the only collective of the hot path (the
weight-blob broadcast) and the utterance sharding.  The data path itself has no collective."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    from radae_amd.engine import DEFAULT_BLOB
    from radae_amd.parallel import broadcast_blob, gather_stats, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = broadcast_blob(DEFAULT_BLOB if rank == 0 else None, torch.device("cpu"), world)
    lo, hi = shard_range(2048, rank, world)
    stats = gather_stats(np.array([hi - lo, float(rank + 1), len(blob)]), torch.device("cpu"), world)
    import hashlib
    q.put((rank, len(blob), hashlib.sha256(blob).hexdigest(), lo, hi, stats.tolist()))
    dist.destroy_process_group()


def test_blob_broadcast_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    ref = open(os.path.join(REPO, "weights", "model19_check3.bin"), "rb").read()
    import hashlib
    assert res[0][1] == res[1][1] == len(ref)
    assert res[0][2] == res[1][2] == hashlib.sha256(ref).hexdigest()      # the bytes, not just their count
    assert (res[0][3], res[0][4], res[1][3], res[1][4]) == (0, 1024, 1024, 2048)
    assert res[0][5] == res[1][5] == [2048.0, 3.0, 2.0 * len(ref)]


def test_config4_world8_shards_blob_and_wait_policy():
    """BASELINE.json configs[3] as far as it can run without eight GPUs: eight gloo ranks, 2048 utterances -> [256 r, 256 r + 256) each, the weight blob
    arrives byte-identical on every rank (sha256), the statistics all-reduce sums over eight ranks, and the host wait policy of such a job -- eight ranks x three
    engines under a 16-core quota -- is the sleeping one on every rank (bench.sync_peers -> $RADE_SYNC_PEERS -> rade_sync_policy)."""
    import hashlib
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=300) for _ in range(8))
    [p.join(timeout=60) for p in procs]
    ref = open(os.path.join(REPO, "weights", "model19_check3.bin"), "rb").read()
    sha = hashlib.sha256(ref).hexdigest()
    for r, (rank, n, h, lo, hi, stats) in enumerate(res):
        assert rank == r and n == len(ref) and h == sha and (lo, hi) == (256 * r, 256 * r + 256)
        assert stats == [2048.0, 36.0, 8.0 * len(ref)]
    sys.path.insert(0, REPO)
    import ctypes as C
    import bench
    from radae_amd import engine
    lib = engine.load_library()
    lib.rade_sync_policy.argtypes = [C.c_int, C.c_double]
    for r in range(8):
        peers = int(bench.sync_peers(8, {"LOCAL_WORLD_SIZE": "8", "LOCAL_RANK": str(r)}))
        assert peers == 8 and lib.rade_sync_policy(3 * peers, 16.0) == 1        # 24 engines > 16 CPUs: rade_batch_rx sleeps
    assert lib.rade_sync_policy(3 * 2, 16.0) == 0                                # two GPUs on the same node: spinning is still right


def test_blob_bytes_identical_after_broadcast():
    # rank-0 bytes parse to the same model as the file (the engine consumes bytes, not a path)
    sys.path.insert(0, REPO)
    import torch
    from radae_amd.parallel import broadcast_blob, shard_range
    from radae_amd.engine import DEFAULT_BLOB
    b = broadcast_blob(DEFAULT_BLOB, torch.device("cpu"), 1)
    assert b == open(DEFAULT_BLOB, "rb").read()
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]      # balanced: no trailing rank is left with (next to) nothing
