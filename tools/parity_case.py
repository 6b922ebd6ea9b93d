#!/usr/bin/env python3
"""One case of tools/parity_sweep.py in detail: parity_case.py <seed> <chan> <EbNo> <fo>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import multipath_g, synth_features
from oracle import oracle_py as O
seed, chan, eb, fo = int(sys.argv[1]), sys.argv[2], float(sys.argv[3]), float(sys.argv[4]); n_mf = 24
O.build(); m = O.Model()
r2 = np.random.default_rng(seed)
feats = synth_features(seed, n_mf * 12); n_sig = n_mf * 960
G = multipath_g(chan, 8000, n_sig, seed + 1) if chan != "awgn" else None
n_pre = int(r2.integers(1000, 9000)); n_tot = n_pre + n_sig + 2304
noise = ((r2.standard_normal(n_tot) + 1j * r2.standard_normal(n_tot)) / np.sqrt(2)).astype(np.complex64)
sigma = sigma_from_EbNodB(eb)
tx = O.Tx(m)
sig = np.concatenate([tx.frame(feats[12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)])
r, fin = O.channel(sig, G, noise[n_pre:n_pre + n_sig], sigma, fo)
e = O.channel_eoo(tx.eoo(), noise[n_pre + n_sig:n_pre + n_sig + 1152], sigma, fo, 0.0, fin)
full = np.concatenate([sigma * noise[:n_pre], r, e, sigma * noise[-1152:]]).astype(np.complex64)
d = O.run_rx_stream(m, full)
eng = BatchEngine(1, max_tx_mf=1, rx_trace_calls=64)
fo_dev, st, _ = eng.rx(torch.tensor(full[None], device="cuda"))
t = eng.rx_trace(0); nv = st[0].n_valid
f = fo_dev.cpu().numpy()[0, :nv]; g = d["features_out"]
print("frames", nv, "feature rms per frame:", np.round(np.sqrt(np.mean((f - g) ** 2, axis=1)), 6))
zv = [i for i in range(len(d["ret"])) if d["ret"][i] & 1]
zg = d["z_hat"]; zt = t["z_hat"]
print("z_hat shapes", np.shape(zg), np.shape(zt))
n = min(len(zg), len(zt))
print("z_hat rms per call:", np.round(np.sqrt(np.mean((np.asarray(zt[:n]) - np.asarray(zg[:n])) ** 2, axis=-1)), 7))
print("z_hat max abs per call (oracle):", np.round(np.abs(np.asarray(zg[:n])).max(axis=-1), 2))
print("fmax diff", np.abs(t["fmax"] - d["fmax"]).max())
