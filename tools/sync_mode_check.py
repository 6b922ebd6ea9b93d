#!/usr/bin/env python3
"""developer aid / test helper (tests/test_hip_parity.py::test_blocking_wait_equals_spinning): three engines on three HIP streams and host threads (bench.py's
pipeline) run two steps each of a small batch; prints one JSON line with the sha256 of everything the receivers returned and how the rade_batch_rx waits were
taken.  The wait policy is read once per process ($RADE_SYNC=spin|block, $RADE_SYNC_PEERS; rade_engine.c: sync_blocking_now), hence a script of its own."""
import hashlib, json, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import synth_features, multipath_g
B, T, depth = int(os.environ.get("SMC_STREAMS", "24")), int(os.environ.get("SMC_FRAMES", "240")), 3
NSTEP = int(os.environ.get("SMC_STEPS", "2"))
n_mf = T // 12
dev = torch.device("cuda")
feats = torch.tensor(np.stack([synth_features(1000 + b, T) for b in range(B)]), device=dev)
G = torch.tensor(np.stack([multipath_g("mpp", 8000, n_mf * 960, 5000 + b) for b in range(B)]), device=dev)
engs = [BatchEngine(B, max_tx_mf=n_mf) for _ in range(depth)]
lanes = [torch.cuda.Stream(device=dev) for _ in range(depth)]
res = [[] for _ in range(depth)]; detail = [[] for _ in range(depth)]
def lane(i):
    with torch.cuda.stream(lanes[i]):
        for k in range(NSTEP):
            e = engs[i]; e.reset()
            zh = iqh = None
            if os.environ.get("SMC_TWOPASS"):
                iq, z = e.tx(feats, want_z=True)
                rx = e.channel(iq, sigma_from_EbNodB(3.0), -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=1 + i + depth * k)
                zh = [hashlib.sha256(z[b].cpu().numpy().tobytes()).hexdigest()[:8] for b in range(B)]
                iqh = [hashlib.sha256(iq[b].cpu().numpy().tobytes()).hexdigest()[:8] for b in range(B)]
            else:
                if os.environ.get("SMC_DUMP"):
                    rx, iq = e.tx_channel(feats, sigma_from_EbNodB(3.0), -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=1 + i + depth * k, want_iq=True)
                    iqh = [hashlib.sha256(iq[b].cpu().numpy().tobytes()).hexdigest()[:8] for b in range(B)]
                else:
                    rx = e.tx_channel(feats, sigma_from_EbNodB(3.0), -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=1 + i + depth * k)
            fo, st, eoo = e.rx(rx)
            torch.cuda.current_stream().synchronize()
            nv = [s.n_valid for s in st]
            per = [hashlib.sha256(fo[b, :nv[b]].cpu().numpy().tobytes()).hexdigest()[:8] for b in range(B)] if os.environ.get("SMC_DETAIL") else None
            rxh = hashlib.sha256(rx.cpu().numpy().tobytes()).hexdigest()[:12] if os.environ.get("SMC_DETAIL") else None
            rxs = [hashlib.sha256(rx[b].cpu().numpy().tobytes()).hexdigest()[:8] for b in range(B)] if os.environ.get("SMC_DETAIL") else None
            detail[i].append((rxh, per, zh, iqh, rxs))
            if os.environ.get("SMC_DUMP"):
                np.savez(os.environ["SMC_DUMP"] + f"_l{i}s{k}.npz", rx=rx.cpu().numpy(), iq=(iq.cpu().numpy() if iqh is not None else np.zeros(1)), z=(z.cpu().numpy() if zh is not None else np.zeros(1)))
            res[i].append((np.concatenate([fo[b, :nv[b]].cpu().numpy().ravel() for b in range(B)]).tobytes(), [(s.n_valid, s.n_calls, s.has_eoo, s.nin, s.sync) for s in st]))
ths = [threading.Thread(target=lane, args=(i,)) for i in range(depth)]
[t.start() for t in ths]; [t.join() for t in ths]
h = hashlib.sha256(); parts = []
for i in range(depth):
    for raw, stl in res[i]:
        h.update(raw); h.update(repr(stl).encode())
        parts.append([hashlib.sha256(raw).hexdigest()[:12], hashlib.sha256(repr(stl).encode()).hexdigest()[:12]])
sc = [e.sync_counts() for e in engs]
print(json.dumps({"sha256": h.hexdigest(), "parts": parts, "detail": detail if os.environ.get("SMC_DETAIL") else None, "rx_waits_blocking": sum(a for a, _ in sc), "rx_waits_spinning": sum(b for _, b in sc),
                  "decoded": int(sum(s[0] for i in range(depth) for _, stl in res[i] for s in stl)), "RADE_SYNC": os.environ.get("RADE_SYNC", "auto"),
                  "RADE_SYNC_PEERS": os.environ.get("RADE_SYNC_PEERS", "1")}))
