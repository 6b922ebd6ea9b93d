#!/usr/bin/env python3
"""developer aid: N launches of the receiver kernel alone on a bench-like batch (one utterance of T feature frames per stream through the
MPP channel) -- the thing to put under rocprofv3 (PC sampling, counters) when only the receiver kernel is of interest.
usage: rx_only.py [launches] [variant 1|2] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import synth_features, multipath_g
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
T = 504; n_mf = T // 12
dev = torch.device("cuda")
base = np.stack([synth_features(3000 + b, T) for b in range(16)])
feats = torch.tensor(base[np.arange(B) % 16], device=dev)
G = torch.empty((B, n_mf * 960, 2), dtype=torch.complex64, device=dev)
for b in range(16):
    G[b::16] = torch.from_numpy(multipath_g("mpp", 8000, n_mf * 960, 7000 + b)).to(dev)
e = BatchEngine(B, max_tx_mf=n_mf)
rx = e.tx_channel(feats, sigma_from_EbNodB(3.0), -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=1)
e.rx(rx); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    e.rx_reset(); f, s, _ = e.rx(rx)
torch.cuda.synchronize()
print("variant", variant, "B", B, "ms/launch", (time.perf_counter() - t0) * 1e3 / n, "decoded frames mean", float(np.mean([x.n_valid for x in s])))
