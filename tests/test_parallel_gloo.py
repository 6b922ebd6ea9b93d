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


def test_blob_bytes_identical_after_broadcast():
    # rank-0 bytes parse to the same model as the file (the engine consumes bytes, not a path)
    sys.path.insert(0, REPO)
    import torch
    from radae_amd.parallel import broadcast_blob, shard_range
    from radae_amd.engine import DEFAULT_BLOB
    b = broadcast_blob(DEFAULT_BLOB, torch.device("cpu"), 1)
    assert b == open(DEFAULT_BLOB, "rb").read()
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]      # balanced: no trailing rank is left with (next to) nothing
