"""Experiment: batches in flight on several engines / HIP streams, one host thread each (rade_batch_rx synchronises its stream)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.channel_tools import multipath_g, synth_features
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
B, T = 256, 1008; n_mf = T // 12; n_sig = n_mf * 960
dev = torch.device('cuda')
feats = torch.tensor(np.stack([synth_features(1000 + b, T) for b in range(B)]), device=dev)
G = torch.empty((B, n_sig, 2), dtype=torch.complex64, device=dev)
for b in range(B): G[b] = torch.from_numpy(multipath_g("mpp", 8000, n_sig, 5000 + b)).to(dev)
sigma = sigma_from_EbNodB(3.0)
for depth in (1, 2, 3):
    engs = [BatchEngine(B, max_tx_mf=n_mf) for _ in range(depth)]
    streams = [torch.cuda.Stream() for _ in range(depth)]
    K = 40
    def worker(i, n, base):
        torch.cuda.set_device(0)
        e = engs[i]
        with torch.cuda.stream(streams[i]):
            for k in range(n):
                e.reset(); iq = e.tx(feats)
                rx = e.channel(iq, sigma, -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=base + k * depth + i)
                e.rx(rx)
    ths = [threading.Thread(target=worker, args=(i, 4, 100)) for i in range(depth)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(i, K // depth, 1)) for i in range(depth)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = (K // depth) * depth
    print(depth, "engines:", round(1e3 * dt / n, 3), "ms/step", round(B * T * n / dt / 1e6, 2), "M frames/s", flush=True)
    for e in engs: e.close()
