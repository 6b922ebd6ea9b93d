"""Per-stream call mix of the bench workload: how many do_radae_rx calls run in search / candidate / sync state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import synth_features, multipath_g
B, T = 256, 1008; n_mf = T // 12
eng = BatchEngine(B, max_tx_mf=n_mf, rx_trace_calls=128)
dev = torch.device('cuda')
feats = torch.tensor(np.stack([synth_features(1000 + b, T) for b in range(B)]), device=dev)
G = torch.empty((B, n_mf * 960, 2), dtype=torch.complex64, device=dev)
for b in range(B):
    G[b] = torch.from_numpy(multipath_g("mpp", 8000, n_mf * 960, 5000 + b)).to(dev)
iq = eng.tx(feats)
rx = eng.channel(iq, sigma_from_EbNodB(3.0), -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=1)
eng.profile(True)
fo, st, _ = eng.rx(rx); torch.cuda.synchronize()
eng.profile(False); pr = eng.profile_get()
print({k: (round(v["ms"], 3), v["launches"]) for k, v in pr.items()})
ns, nc, ny, nf = [], [], [], []
for b in range(B):
    t = eng.rx_trace(b)
    sb, sa = t["state_before"], t["state_after"]
    ns.append(int((sb == 0).sum())); nc.append(int((sb == 1).sum())); ny.append(int((sb == 2).sum())); nf.append(int(((sb == 2) & (sa == 0)).sum()))
ns, nc, ny, nf = map(np.array, (ns, nc, ny, nf))
for name, a in (("search", ns), ("candidate", nc), ("sync", ny), ("sync losses", nf)):
    print(f"{name:12s} min {a.min():3d} mean {a.mean():6.2f} max {a.max():3d}")
cost = (ns + nc) * 3.0 + ny
print("cost units (search=3, sync=1): mean", cost.mean(), "max", cost.max(), "argmax", int(cost.argmax()))
print("hist of sync losses", np.bincount(nf))
