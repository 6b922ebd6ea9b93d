#!/usr/bin/env python3
"""developer aid: the two receiver kernels on the SAME received samples (bench-like workload, 512 different streams = two workgroups
per CU for k_rx_sync2): per-stream decoded-frame counts, per-call traces (first 96 calls) and features must agree -- discrete outputs
exactly (except refine() near-ties, counted), features to 1e-5 RMS -- and k_rx_sync2 must reproduce itself bit for bit."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import synth_features, multipath_g
B = int(os.environ.get("STRESS_B", "512")); T = int(os.environ.get("STRESS_T", "504")); n_mf = T // 12
os.environ.pop("RADE_RX_VARIANT", None)
dev = torch.device("cuda")
feats = torch.tensor(np.stack([synth_features(3000 + b, T) for b in range(B)]), device=dev)
G = torch.empty((B, n_mf * 960, 2), dtype=torch.complex64, device=dev)
for b in range(B):
    G[b] = torch.from_numpy(multipath_g("mpp", 8000, n_mf * 960, 7000 + b)).to(dev)
INT_KEYS = ["state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]
out = {"streams": B, "frames": T, "seeds": {}}
e1 = BatchEngine(B, max_tx_mf=n_mf, rx_trace_calls=96); e2 = BatchEngine(B, max_tx_mf=n_mf, rx_trace_calls=96, flags=0x200)
for seed in (1, 2, 3):
    e1.reset(); e2.reset()
    rx = e1.tx_channel(feats, sigma_from_EbNodB(3.0), -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=seed)
    f1, s1, _ = e1.rx(rx); f2, s2, _ = e2.rx(rx)
    e2.rx_reset(); f2b, s2b, _ = e2.rx(rx)
    f1, f2, f2b = f1.cpu().numpy(), f2.cpu().numpy(), f2b.cpu().numpy()
    bad, ties, worst, selfdiff = [], 0, 0.0, 0
    for b in range(B):
        if s2[b].n_valid != s2b[b].n_valid or not np.array_equal(f2[b], f2b[b]): selfdiff += 1
        t1, t2 = e1.rx_trace(b), e2.rx_trace(b)
        same = all(np.array_equal(t1[k], t2[k]) for k in INT_KEYS) and s1[b].n_valid == s2[b].n_valid
        if not same:
            dfm = np.abs(t1["fmax"][:min(len(t1["fmax"]), len(t2["fmax"]))] - t2["fmax"][:min(len(t1["fmax"]), len(t2["fmax"]))])
            if len(dfm) and 1e-9 < dfm.max() < 0.0501: ties += 1
            else: bad.append(b)
            continue
        nv = s1[b].n_valid
        if nv: worst = max(worst, float(np.sqrt(np.mean((f1[b, :nv] - f2[b, :nv]) ** 2))))
    out["seeds"][seed] = {"streams_discrete_mismatch": bad[:10], "n_mismatch": len(bad), "refine_near_tie_streams": ties, "feat_rms_max_between_kernels": worst,
                          "rx2_rerun_streams_not_bitwise_equal": selfdiff, "decoded_frames_mean": float(np.mean([s.n_valid for s in s2]))}
print(json.dumps(out))
