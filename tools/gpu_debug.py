#!/usr/bin/env python3
"""Diagnostic run on a GPU box: HIP engine vs golden fixtures and the CPU oracle, printed (not asserted)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from oracle import oracle_py as O

G = os.path.join(REPO, "tests", "golden")
def cmp(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape: print(f"{name}: SHAPE {a.shape} vs {b.shape}"); return
    if a.size == 0: print(f"{name}: empty"); return
    d = np.abs(a.astype(np.complex128) - b.astype(np.complex128))
    print(f"{name:28s} max|d|={d.max():.3e} rms={np.sqrt((d**2).mean()):.3e} scale={np.abs(b).max():.3g} nan={int(np.isnan(np.abs(a)).sum())}")

dev = torch.device("cuda")
print(torch.cuda.get_device_name(0))
e = np.load(os.path.join(G, "enc_tx.npz"))
eng = BatchEngine(2, max_tx_mf=10)
f = torch.tensor(e["features"], device=dev)
iq, z = eng.tx(f, want_z=True); torch.cuda.synchronize()
cmp("enc z (whole)", z.cpu().numpy(), e["z"]); cmp("tx (whole)", iq.cpu().numpy().reshape(2, 10, 960), e["tx"])
eng.tx_reset()
outs, zs = [], []
for k in range(10):
    a, b = eng.tx(f[:, 12*k:12*k+12].contiguous(), want_z=True); outs.append(a); zs.append(b)
torch.cuda.synchronize()
cmp("enc z (streamed)", torch.cat(zs, 1).cpu().numpy(), e["z"]); cmp("tx (streamed)", torch.cat(outs, 1).cpu().numpy().reshape(2, 10, 960), e["tx"])
c = np.load(os.path.join(G, "consts.npz"))
cmp("eoo default", eng.tx_eoo().cpu().numpy()[0], c["eoo_default"])
eng.set_eoo_bits(np.stack([c["eoo_bits_in"]] * 2)); cmp("eoo bits", eng.tx_eoo().cpu().numpy()[1], c["eoo_with_bits"])
eng.close()

m = O.Model()
for name in ["mpp", "awgn"]:
    g = np.load(os.path.join(G, f"chan_{name}.npz"))
    eng = BatchEngine(1, max_tx_mf=1, rx_trace_calls=64)
    sigma = float(g["sigma"]); n_sig = len(g["tx"]); n_pre = len(g["noise_pre"]); n_post = len(g["noise_post"])
    noise = np.concatenate([g["noise_pre"].astype(np.complex64), g["noise"], g["noise_eoo"], g["noise_post"].astype(np.complex64)])
    rx = eng.channel(torch.tensor(g["tx"][None], device=dev), sigma, float(g["freq_offset"]), n_pre, n_post, True,
                     G=torch.tensor(g["G"][None], device=dev), noise=torch.tensor(noise[None], device=dev))
    torch.cuda.synchronize()
    cmp(f"chan {name} rx_full", rx.cpu().numpy()[0], g["rx_full"])
    eng.close()

for name in ["awgn", "mpp", "slip_plus", "slip_minus", "foff"]:
    g = np.load(os.path.join(G, f"rxtrace_{name}.npz"))
    eng = BatchEngine(1, max_tx_mf=1, rx_trace_calls=64, flags=4 if name == "foff" else 0)
    x = torch.tensor(g["rx_in"][None], device=dev)
    t0 = time.time()
    feats, st, eoo = eng.rx(x); torch.cuda.synchronize()
    dt = time.time() - t0
    d = eng.rx_trace(0)
    print(f"== {name}: calls {st[0].n_calls} valid {st[0].n_valid} consumed {st[0].consumed} in {dt*1e3:.1f} ms")
    for k in ["state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]:
        if d[k].shape != g[k].shape or not np.array_equal(d[k], g[k]):
            print("   INT MISMATCH", k); print("    got", d[k].tolist()); print("    ref", g[k].tolist())
    for k in ["fmax", "Dthresh", "Dtmax12", "Dtmax12_eoo", "snrdB_3k_est"]:
        cmp("   " + k, d[k], g[k])
    cmp("   z_hat", d["z_hat"], g["z_hat"])
    nv = st[0].n_valid
    cmp("   features", feats.cpu().numpy()[0, :nv], g["features_out"])
    cmp("   eoo", d["eoo_out"], g["eoo_out"])
    eng.close()
print("done")
