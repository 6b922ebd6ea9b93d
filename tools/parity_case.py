#!/usr/bin/env python3
"""Developer aid (GPU box; uses the oracle as the checker like the tests do): one case of tools/parity_sweep.py again, with the per-call traces of the
HIP receiver and the oracle side by side from the first differing call on, and the oracle's refine() arg-max margin (runner-up cell relative to the
winner) at every call.   usage: SWEEP_SEED=<seed of the sweep> python tools/parity_case.py <case index> [<case index> ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import multipath_g, synth_features
from oracle import oracle_py as O

n_mf = 24
O.build(); m = O.Model()
want = sorted(int(a) for a in sys.argv[1:])
rng = np.random.default_rng(int(os.environ.get("SWEEP_SEED", "2026")))
for case in range(max(want) + 1):
    seed = int(rng.integers(1, 1 << 30)); eb = float(rng.uniform(-1.0, 12.0)); fo = float(rng.uniform(-40.0, 40.0))
    chan = ["awgn", "mpp", "mpd", "mpg"][int(rng.integers(0, 4))]
    if case not in want:
        continue
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
    # the oracle call by call, with the refine() margin after each call
    rx = O.Rx(m); pos = 0; rows = []
    bpf = O.Bpf(); rx_buf = np.zeros(2112, np.complex64); bufs = []; pil = O.get_const("p", 320, True).astype(np.complex128)
    mg = C.c_double.in_dll(O.lib(), "orc_debug_refine_margin"); st_ = C.c_int.in_dll(O.lib(), "orc_debug_refine_second_t"); sf_ = C.c_double.in_dll(O.lib(), "orc_debug_refine_second_f")
    while pos + rx.nin() <= len(full):
        nin = rx.nin(); sb = rx.trace().state; mg.value = 1.0
        filt = bpf.run(full[pos:pos + nin]); rx_buf = np.concatenate([rx_buf[nin:], filt]).astype(np.complex64); bufs.append((rx_buf.copy(), rx.trace().tmax, rx.trace().fmax))
        ret, _, _, _ = rx.frame(full[pos:pos + nin]); pos += nin
        t = rx.trace()
        rows.append(dict(sb=sb, sa=t.state, nin=nin, ret=ret, tmax=t.tmax, fmax=t.fmax, D=t.Dtmax12, Dth=t.Dthresh, vc=t.valid_count, uw=t.uw_errors, margin=mg.value, sec_t=st_.value, sec_f=sf_.value))
    eng = BatchEngine(1, max_tx_mf=1, rx_trace_calls=64)
    eng.rx(torch.tensor(full[None], device="cuda"))
    g = eng.rx_trace(0)
    n = min(len(rows), len(g["tmax"]))
    first = next((i for i in range(n) if rows[i]["tmax"] != g["tmax"][i] or abs(rows[i]["fmax"] - g["fmax"][i]) > 1e-9 or rows[i]["sa"] != g["state_after"][i] or rows[i]["ret"] != g["ret"][i]), None)
    print(f"case {case}: seed {seed} {chan} Eb/No {eb:.3f} dB fo {fo:.3f} Hz, {len(rows)} oracle calls / {len(g['tmax'])} device calls, first differing call {first}")
    if first is None:
        continue
    for i in range(max(0, first - 2), min(n, first + 6)):
        o = rows[i]
        print(f"  call {i:2d} oracle: state {o['sb']}->{o['sa']} ret {o['ret']} tmax {o['tmax']:4d} fmax {o['fmax']:9.4f} D {o['D']:.7g} Dth {o['Dth']:.7g} vc {o['vc']} uw {o['uw']}"
              f" | refine margin {o['margin']:.3e} (runner-up t {o['sec_t']} f {o['sec_f']:.4f})")
        print(f"          device: state {g['state_before'][i]}->{g['state_after'][i]} ret {g['ret'][i]} tmax {g['tmax'][i]:4d} fmax {g['fmax'][i]:9.4f} D {g['Dtmax12'][i]:.7g} Dth {g['Dthresh'][i]:.7g}"
              f" vc {g['valid_count'][i]} uw {g['uw_errors'][i]}")
    eng.close()
    # the refine() surface of the first differing call recomputed here (numpy complex128 sums -> complex64 -> float32 magnitudes, dsp.py:233-270)
    if rows[first]["sb"] == 2:
        buf, tm, fm = bufs[first]
        fr = np.arange(fm - 1, fm + 1, 0.1); t0 = max(tm - 8, 0); n = np.arange(160)
        surf = np.zeros((len(fr), 16), np.float32)
        for fi, f in enumerate(fr):
            w = 2 * np.pi * f / 8000.0
            wp1 = np.exp(-1j * w * n) * np.conj(pil); wp2 = wp1 * np.exp(-1j * w * 960)
            for ti in range(16):
                a = np.complex64(np.sum(buf[t0 + ti:t0 + ti + 160].astype(np.complex128) * wp1)); b_ = np.complex64(np.sum(buf[t0 + ti + 960:t0 + ti + 1120].astype(np.complex128) * wp2))
                surf[fi, ti] = np.abs(np.complex64(a + b_))
        order = np.argsort(-surf.ravel())[:6]
        print(f"  numpy surface of call {first}: {len(fr)} frequencies from {fr[0]:.4f} (start tmax {tm}, fmax {fm:.4f}); top cells:")
        for o in order:
            fi, ti = divmod(int(o), 16)
            print(f"    t {t0 + ti} f index {fi} ({fr[fi]:.4f}) |Dt| {surf[fi, ti]:.7g}   -> fmax would be {0.9 * fm + 0.1 * fr[fi]:.4f}")
