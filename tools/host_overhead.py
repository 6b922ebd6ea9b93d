"""Host-side time of each engine call in the bench loop (the GPU work is asynchronous except inside rx)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import synth_features, multipath_g
B, T = 256, 1008; n_mf = T // 12
eng = BatchEngine(B, max_tx_mf=n_mf); dev = torch.device("cuda")
feats = torch.tensor(np.stack([synth_features(1000 + b, T) for b in range(B)]), device=dev)
G = torch.empty((B, n_mf * 960, 2), dtype=torch.complex64, device=dev)
for b in range(B): G[b] = torch.from_numpy(multipath_g("mpp", 8000, n_mf * 960, 5000 + b)).to(dev)
sigma = sigma_from_EbNodB(3.0)
acc = {"reset": 0.0, "tx": 0.0, "channel": 0.0, "rx": 0.0}; n = 20
for k in range(n + 3):
    if k == 3: acc = {key: 0.0 for key in acc}; torch.cuda.synchronize(); t_all = time.perf_counter()
    t = time.perf_counter(); eng.reset(); acc["reset"] += time.perf_counter() - t
    t = time.perf_counter(); iq = eng.tx(feats); acc["tx"] += time.perf_counter() - t
    t = time.perf_counter(); rx = eng.channel(iq, sigma, -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=k); acc["channel"] += time.perf_counter() - t
    t = time.perf_counter(); out = eng.rx(rx); acc["rx"] += time.perf_counter() - t
torch.cuda.synchronize(); tot = time.perf_counter() - t_all
print({k: round(1e3 * v / n, 3) for k, v in acc.items()}, "ms host per call; total per step", round(1e3 * tot / n, 3))
