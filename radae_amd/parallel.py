"""Multi-GPU plumbing.  The hot path shards by utterance (streams are independent), so the
only collective is a one-off broadcast of the DNNw weight blob from rank 0 (RCCL over xGMI when the
process group backend is "nccl"; gloo in the CPU tests).  One process per GPU."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def broadcast_blob(path_on_rank0: Optional[str], device, world_size: int) -> bytes:
    """Rank 0 reads the blob; every rank returns identical bytes.  world_size == 1: plain file read."""
    import torch
    if world_size <= 1:
        with open(path_on_rank0, "rb") as f:
            return f.read()
    import torch.distributed as dist
    rank = dist.get_rank()
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == 0:
        with open(path_on_rank0, "rb") as f:
            raw = f.read()
        n[0] = len(raw)
    dist.broadcast(n, src=0)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    if rank == 0:
        buf.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    dist.broadcast(buf, src=0)
    return buf.cpu().numpy().tobytes()


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous balanced utterance shard [lo, hi) of rank `rank`: n // world each, the first n % world ranks one more (same rule
    as rade_multi_shard in C; config 4: GPU g gets u in [256g, 256g+256))."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_stats(local: np.ndarray, device, world_size: int) -> np.ndarray:
    """Sum a small vector of per-rank statistics (frames, sum loss, bit errors) over all ranks."""
    if world_size <= 1:
        return np.asarray(local, dtype=np.float64)
    import torch
    import torch.distributed as dist
    t = torch.tensor(np.asarray(local, dtype=np.float64), device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
