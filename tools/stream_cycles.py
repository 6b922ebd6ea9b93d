#!/usr/bin/env python3
"""Per-stream duration of the receiver launch on the bench workload (shader-clock cycles of each stream's workgroup): the launch
lasts as long as its slowest stream, so max / mean is the share of the chip that idles in the tail.  Prints one JSON object."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import synth_features, multipath_g
B, T = 256, 1008; n_mf = T // 12
eng = BatchEngine(B, max_tx_mf=n_mf)
dev = torch.device("cuda")
feats = torch.tensor(np.stack([synth_features(1000 + b, T) for b in range(B)]), device=dev)
G = torch.empty((B, n_mf * 960, 2), dtype=torch.complex64, device=dev)
for b in range(B):
    G[b] = torch.from_numpy(multipath_g("mpp", 8000, n_mf * 960, 5000 + b)).to(dev)
out = {"workload": "bench.py step (256 streams x 1008 frames, MPP 3 dB, -11 Hz)", "seeds": {}}
for seed in (1, 2, 3):
    eng.reset()
    rx = eng.channel(eng.tx(feats), sigma_from_EbNodB(3.0), -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=seed)
    eng.profile(True); fo, st, _ = eng.rx(rx); torch.cuda.synchronize(); eng.profile(False)
    ms = eng.profile_get()["rx_sync"]["ms"]
    c = eng.rx_stream_cycles().astype(np.float64)
    calls = np.array([s.n_calls for s in st]); sync = np.array([s.n_valid + s.has_eoo for s in st])
    srch = calls - sync
    # least-squares cost per call type: cycles ~ a * search_calls + b * sync_calls
    A = np.stack([srch, sync], 1).astype(np.float64); coef = np.linalg.lstsq(A, c, rcond=None)[0]
    out["seeds"][str(seed)] = {"kernel_ms": ms, "cycles_min": c.min(), "cycles_p10": float(np.percentile(c, 10)), "cycles_median": float(np.median(c)),
                               "cycles_p90": float(np.percentile(c, 90)), "cycles_max": c.max(), "cycles_mean": c.mean(), "max_over_mean": c.max() / c.mean(),
                               "implied_clock_GHz": c.max() / (ms * 1e-3) / 1e9, "search_calls_min_max": [int(srch.min()), int(srch.max())],
                               "cycles_per_search_call_fit": coef[0], "cycles_per_sync_call_fit": coef[1], "slowest_stream": int(c.argmax()),
                               "slowest_stream_calls": [int(srch[c.argmax()]), int(sync[c.argmax()])]}
print(json.dumps(out))
