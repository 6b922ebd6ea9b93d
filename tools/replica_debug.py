"""developer aid: B copies of one golden receive stream through the engine; which replicas differ from replica 0, and at which call / field first"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine
name = sys.argv[1] if len(sys.argv) > 1 else "mpp"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"rxtrace_{name}.npz"))
buf = torch.tensor(np.stack([g["rx_in"]] * B), device="cuda")
eng = BatchEngine(B, max_tx_mf=1, rx_trace_calls=64)
dbg = None
if hasattr(eng.lib, "rd_debug_set_buf"):
    import ctypes
    dbg = torch.zeros((B, 32, 1120), dtype=torch.complex64, device="cuda")
    eng.lib.rd_debug_set_buf.argtypes = [ctypes.c_void_p]; eng.lib.rd_debug_set_buf(dbg.data_ptr())
for rep in range(2):
    eng.rx_reset()
    f, st, _ = eng.rx(buf); f = f.cpu().numpy()
    t0 = eng.rx_trace(0)
    bad = []
    for b in range(1, B):
        t = eng.rx_trace(b)
        first = None
        for k in ["Dtmax12_eoo", "snrdB_3k_est", "Dtmax12"] + [kk for kk in t0 if kk not in ("Dtmax12", "Dtmax12_eoo", "snrdB_3k_est", "z_hat", "eoo_out")]:
            if t[k].shape != t0[k].shape: first = (k, "shape"); break
            d = np.nonzero(np.atleast_1d((t[k] != t0[k])).reshape(len(t0[k]), -1).any(1))[0] if len(t0[k]) else []
            if len(d) and (first is None or d[0] < first[1]): first = (k, int(d[0]), np.asarray(t[k][d[0]]).ravel()[:3].tolist(), np.asarray(t0[k][d[0]]).ravel()[:3].tolist())
        featdiff = not np.array_equal(f[b], f[0])
        fd = np.nonzero((f[b] != f[0]).any(1))[0]
        if first or featdiff: bad.append((b, first, "feat frames differing:", fd[:6].tolist(), len(fd), (st[b].n_calls, st[b].n_valid)))
    print(f"rep {rep}: {len(bad)} of {B - 1} replicas differ from replica 0 (calls {st[0].n_calls}, valid {st[0].n_valid})")
    for x in bad[:12]: print("   ", x)
    if dbg is not None:
        d = dbg.cpu().numpy()
        for x in bad[:10]:
            bb = x[0]
            diff = np.nonzero((d[bb] != d[0]).any(1))[0]
            if len(diff):
                c = diff[0]; idx = np.nonzero(d[bb, c] != d[0, c])[0]
                i0 = idx[0]; w = d[bb, c, i0]; r = d[0, c, i0]
                hits = [(cc, int(ii)) for cc in range(max(0, c - 3), c + 1) for ii in np.nonzero(d[0, cc] == w)[0][:3]]
                hre = [(cc, int(ii)) for cc in range(max(0, c - 3), c + 1) for ii in np.nonzero(d[0, cc].real == w.real)[0][:3]]
                him = [(cc, int(ii)) for cc in range(max(0, c - 3), c + 1) for ii in np.nonzero(d[0, cc].imag == w.imag)[0][:3]]
                print(f"        sample {i0}: wrong {w} right {r}; wrong value found in good stream at (call, index) {hits}; re-part matches {hre}; im-part matches {him}")
                print(f"      replica {bb}: first call with different filtered samples {c}: {len(idx)} samples differ, indices {idx[:24].tolist()} ... {idx[-3:].tolist()}; threads {sorted(set((idx // 5).tolist()))[:20]}")
    for x in bad[:8]:
        bb = x[0]; t = eng.rx_trace(bb)
        if x[1] and x[1][0] == "z_all":
            r = x[1][1]; d = np.nonzero(t["z_all"][r].ravel() != t0["z_all"][r].ravel())[0]
            el = d // 2
            print(f"      replica {bb} call-row {r}: {len(d)} of 240 floats differ; (symbol, carrier) = {sorted(set((int(e) // 30 + 1, int(e) % 30) for e in el))[:40]}  max rel {np.abs(t['z_all'][r] - t0['z_all'][r]).max() / np.abs(t0['z_all'][r]).max():.2e}")
    # vs golden
    keys = [k for k in ("state_after", "nin_after", "ret", "tmax", "f_ind_max", "uw_errors") if k in g]
    print("   replica 0 vs golden:", {k: bool(np.array_equal(t0[k], g[k][:len(t0[k])])) for k in keys})
