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
