"""Race detector: B identical streams must produce bit-identical traces and features in every slot."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine
B = int(os.environ.get("SLOTS", "256"))
bad = 0
for name in ("rxtrace_awgn", "rxtrace_mpp", "rxtrace_foff"):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name + ".npz"))
    x = g["rx_in"]
    eng = BatchEngine(B, max_tx_mf=1, rx_trace_calls=64, flags=4 if name.endswith("foff") else 0)
    buf = torch.tensor(np.stack([x] * B), device="cuda")
    for rep in range(3):
        eng.rx_reset()
        f, st, _ = eng.rx(buf)
        f = f.cpu().numpy()
        same = [b for b in range(B) if not np.array_equal(f[b], f[0]) or st[b].n_calls != st[0].n_calls or st[b].n_valid != st[0].n_valid]
        t0 = eng.rx_trace(0)
        ok_gold = all(np.array_equal(t0[k], g[k]) for k in ("state_after", "tmax", "uw_errors", "f_ind_max"))
        print(name, "rep", rep, "slots differing from slot 0:", len(same), same[:8], "slot0 == golden:", ok_gold)
        bad += len(same) + (0 if ok_gold else 1)
    eng.close()
print("FAILED" if bad else "all slots identical")

# run-to-run determinism on the bench workload (256 different streams, MPP, 3 dB): two passes must agree bit for bit
from radae_amd.engine import sigma_from_EbNodB
from radae_amd.channel_tools import synth_features, multipath_g
Bb, T = 256, 1008; n_mf = T // 12
eng = BatchEngine(Bb, max_tx_mf=n_mf)
feats = torch.tensor(np.stack([synth_features(1000 + b, T) for b in range(Bb)]), device="cuda")
G = torch.empty((Bb, n_mf * 960, 2), dtype=torch.complex64, device="cuda")
for b in range(Bb): G[b] = torch.from_numpy(multipath_g("mpp", 8000, n_mf * 960, 5000 + b)).to("cuda")
outs = []
for rep in range(3):
    eng.reset()
    iq = eng.tx(feats)
    rx = eng.channel(iq, sigma_from_EbNodB(3.0), -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=7)
    f, st, eoo = eng.rx(rx)
    outs.append((f.cpu().numpy(), [(s.consumed, s.n_calls, s.n_valid, s.has_eoo, s.state) for s in st], eoo.cpu().numpy()))
same = all(np.array_equal(outs[0][0], o[0]) and outs[0][1] == o[1] and np.array_equal(outs[0][2], o[2]) for o in outs[1:])
print("bench workload, 3 runs bit-identical:", same)
if not same: print("FAILED")
