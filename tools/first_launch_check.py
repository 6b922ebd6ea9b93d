#!/usr/bin/env python3
"""Race hunt: 256 streams = 32 replicas of 8 utterances through the receiver as the FIRST GPU work of the process; every replica
must match replica 0 of its utterance call by call.  Prints the first deviation per odd stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import synth_features
B, T = 256, 1008; n_mf = T // 12
base = [synth_features(3000 + u, T) for u in range(8)]
feats = np.stack([base[b % 8] for b in range(B)])
dev = torch.device("cuda")
KEYS = ["state_after", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]
rng = np.random.default_rng(9)
n_tot = 4000 + n_mf * 960 + 1152 + 1152
nz = ((rng.standard_normal((8, n_tot)) + 1j * rng.standard_normal((8, n_tot))) / np.sqrt(2)).astype(np.complex64)
noise = torch.tensor(np.concatenate([nz] * 32), device=dev)
for rep in range(int(os.environ.get("REPS", "3"))):
    TR = int(os.environ.get("TRACE", "100"))
    eng = BatchEngine(B, max_tx_mf=n_mf, rx_trace_calls=TR)
    iq = eng.tx(torch.tensor(feats, device=dev))
    rx = eng.channel(iq, sigma_from_EbNodB(10.0), 11.0, n_pre=4000, n_post=1152, with_eoo=True, noise=noise)
    fo, st, _ = eng.rx(rx)
    iq_bad = [b for b in range(8, B) if not torch.equal(iq[b], iq[b % 8])]; rx_bad = [b for b in range(8, B) if not torch.equal(rx[b], rx[b % 8])]
    if iq_bad or rx_bad: print(f"rep {rep}: tx replicas differ {iq_bad[:8]}, channel replicas differ {rx_bad[:8]}")
    nv = np.array([s.n_valid for s in st])
    bad = 0
    if TR == 0:
        odd = [b for b in range(8, B) if nv[b] != nv[b % 8] or not torch.equal(fo[b], fo[b % 8])]
        for b in odd[:3]:
            d = (fo[b] - fo[b % 8]).abs(); fr = torch.nonzero((fo[b] != fo[b % 8]).any(dim=1)).flatten().cpu().numpy()
            print(f"   stream {b}: differing frames {fr[:20]} (of {len(fr)}), max abs diff {float(d.max()):.3e}, nv {nv[b]} vs {nv[b % 8]}")
        print(f"rep {rep}: odd streams {odd[:10]} nv {[int(nv[b]) for b in odd[:10]]} (first differing frame {[int(torch.nonzero((fo[b] != fo[b % 8]).any(dim=1))[0]) for b in odd[:10] if (fo[b] != fo[b % 8]).any()]})")
        eng.close(); continue
    ref = [eng.rx_trace(u) for u in range(8)]
    for b in range(8, B):
        t = eng.rx_trace(b); r = ref[b % 8]
        dev_keys = {k: int(np.argmax(t[k][:len(r[k])] != r[k][:len(t[k])])) for k in KEYS if len(t[k]) != len(r[k]) or not np.array_equal(t[k], r[k])}
        zdiff = np.abs(t["z_all"][:min(len(t["z_all"]), len(r["z_all"]))] - r["z_all"][:min(len(t["z_all"]), len(r["z_all"]))]).max(axis=1)
        fdiff = (fo[b] != fo[b % 8]).any(dim=1).cpu().numpy()
        if dev_keys or zdiff.max() > 0 or fdiff.any():
            bad += 1
            c0 = min(dev_keys.values()) if dev_keys else -1
            print(f"rep {rep} stream {b} (utt {b % 8}): nv {nv[b]} vs {nv[b % 8]}; first differing call per key {dev_keys}; first z diff call {int(np.argmax(zdiff > 0)) if zdiff.max() > 0 else -1}; first feature-frame diff {int(np.argmax(fdiff)) if fdiff.any() else -1}")
            if c0 >= 0:
                for k in KEYS: print("   ", k, t[k][max(0, c0 - 2):c0 + 3], r[k][max(0, c0 - 2):c0 + 3])
    print(f"rep {rep}: {bad} odd streams, nv min {nv.min()} max {nv.max()}")
    eng.close()
