#!/usr/bin/env python3
"""Throughput of the batched single-carrier modem: B streams x NF frames, tx then rx, inputs resident in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.sc import SingleCarrierBatch
B, NF = int(os.environ.get("SC_STREAMS", "4096")), int(os.environ.get("SC_FRAMES", "100"))
dev = torch.device("cuda")
m = SingleCarrierBatch(B, fcentreHz=1500.0)
sy = torch.sign(torch.randn((B, NF, 80), device=dev))
def step():
    m.reset()
    tx = m.tx(sy)
    rx = tx + 0.05 * torch.view_as_complex(torch.randn((B, tx.shape[1], 2), device=dev))
    return m.rx(rx, max_frames=NF)
step(); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
m.reset(); ev[0].record(); tx = m.tx(sy); ev[1].record()
rx = (tx + 0.05 * torch.view_as_complex(torch.randn((B, tx.shape[1], 2), device=dev))).contiguous(); torch.cuda.synchronize()
ev[2].record(); pay, zh, fr, st = m.rx(rx, max_frames=NF); ev[3].record(); torch.cuda.synchronize()
t_tx, t_rx = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])
nfr = sum(s.n_frames for s in st)
synced = int((fr["state"] == 1).sum())
bytes_tx = B * NF * (80 * 4 + 384 * 8); bytes_rx = nfr * (384 * 8 + 80 * 8 + 80 * 4 + 48)
print(f"streams {B} frames/stream {NF}: tx {t_tx:.3f} ms ({B*NF/t_tx/1e3:.2f} M frames/s, {bytes_tx/t_tx/1e6:.1f} GB/s algorithmic)  "
      f"rx {t_rx:.3f} ms ({nfr/t_rx/1e3:.2f} M frames/s, {bytes_rx/t_rx/1e6:.1f} GB/s algorithmic), synced frames {synced}/{nfr}")
print(f"real-time factor of the receiver: one frame is 40 ms of signal -> {nfr*0.04/(t_rx*1e-3):.0f} x real time")
