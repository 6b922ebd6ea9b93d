#!/usr/bin/env python3
"""Randomised receiver parity sweep (developer aid, uses the oracle as the checker like the tests do): N random channels /
offsets / SNRs; the oracle produces the received samples, the HIP receiver and the oracle receiver both consume exactly those
samples, every per-call discrete output must be equal and the decoded features equal to 1e-4 RMS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import multipath_g, synth_features
from oracle import oracle_py as O
INT_KEYS = ["state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]
N = int(os.environ.get("SWEEP_N", "48")); n_mf = 24
O.build(); m = O.Model()
rng = np.random.default_rng(int(os.environ.get("SWEEP_SEED", "2026")))
bad = 0; ties = 0; dties = 0; tties = 0; sties = 0; tot_calls = 0; tot_valid = 0
for case in range(N):
    seed = int(rng.integers(1, 1 << 30)); eb = float(rng.uniform(-1.0, 12.0)); fo = float(rng.uniform(-40.0, 40.0))
    chan = ["awgn", "mpp", "mpd", "mpg"][int(rng.integers(0, 4))]
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
    t = eng.rx_trace(0)
    nv = st[0].n_valid
    ok = all(np.array_equal(t[k], d[k]) for k in INT_KEYS) and nv == len(d["features_out"])
    if ok and nv:
        ok = float(np.sqrt(np.mean((fo_dev.cpu().numpy()[0, :nv] - d["features_out"]) ** 2))) < 1e-4
    tot_calls += st[0].n_calls; tot_valid += nv
    if not ok:
        first = {k: int(np.argmax(t[k] != d[k])) for k in INT_KEYS if not np.array_equal(t[k], d[k])}
        dfm = float(np.abs(t["fmax"] - d["fmax"]).max()) if len(t["fmax"]) == len(d["fmax"]) else -1.0
        # discrete outputs all equal but fmax moved by a grid step.  Rounds 2-3 read this as a refine() tie (two 0.1 Hz bins equal to within the float rounding of
        # the complex128 sums); it was the device searching a 20-point grid where the reference's np.arange had 21 points (HISTORY.md 4) and has not occurred
        # since that was fixed -- the class stays so that a regression shows up under its old name
        tie = not first and 0.0 < dfm < 0.0501
        if tie:
            # ... and it only counts as a tie if the oracle's own arg-max margin (runner-up cell relative to the winner, float32 magnitudes) at the first call whose
            # fmax differs is within float32 rounding: the oracle again, call by call
            import ctypes as C
            mg = C.c_double.in_dll(O.lib(), "orc_debug_refine_margin")
            i_first = int(np.argmax(t["fmax"] != d["fmax"])); rxo = O.Rx(m); pos = 0; margin = 1.0
            for i in range(i_first + 1):
                nin = rxo.nin(); mg.value = 1.0; rxo.frame(full[pos:pos + nin]); pos += nin; margin = mg.value
            tie = margin < 3e-7
            print(f"  (oracle arg-max margin at call {i_first}: {margin:.3e})")
        # detect_pilots picks the arg-max of |Dt1| + |Dt2| (float32) over 960 x 40 cells: on a noise-only call two cells can be equal to
        # within ONE float32 ulp, and the FFT-convolution correlator and the direct sums round differently; if the only differing
        # outputs are (tmax, f_ind_max) of calls whose maxima agree to 1e-6 and every later output is equal again, it is that tie
        dtie = False
        if first and set(first) <= {"tmax", "f_ind_max"} and len(t["Dtmax12"]) == len(d["Dtmax12"]):
            calls = [i for i in range(len(t["tmax"])) if t["tmax"][i] != d["tmax"][i] or t["f_ind_max"][i] != d["f_ind_max"][i]]
            dtie = all(abs(float(t["Dtmax12"][i]) - float(d["Dtmax12"][i])) <= 1e-6 * abs(float(d["Dtmax12"][i])) and int(d["state_before"][i]) != 2 for i in calls)
        # a refine() tie can also be between two cells of DIFFERENT timing (then tmax, and with it a few later outputs, differ until the estimates meet again):
        # accepted under the same rule -- the oracle's own arg-max margin at the first differing call is within float32 rounding -- if, in addition, both
        # receivers decode the same number of frames and their traces are equal again from some later call on to the end
        ttie = False
        if first and not dtie and len(t["tmax"]) == len(d["tmax"]) and nv == len(d["features_out"]):
            import ctypes as C
            mg = C.c_double.in_dll(O.lib(), "orc_debug_refine_margin")
            i_first = min(first.values()); rxo = O.Rx(m); pos = 0; margin = 1.0
            for i in range(i_first + 1):
                nin = rxo.nin(); mg.value = 1.0; rxo.frame(full[pos:pos + nin]); pos += nin; margin = mg.value
            differ = [i for i in range(len(t["tmax"])) if any(t[k][i] != d[k][i] for k in INT_KEYS)]
            ttie = margin < 3e-7 and int(d["state_before"][i_first]) == 2 and max(differ) < len(t["tmax"]) - 1 and max(differ) - i_first <= 8
            print(f"  (oracle arg-max margin at call {i_first}: {margin:.3e}; calls that differ: {differ})")
        # rade_snrdB_3k_est is the float32 estimate TRUNCATED to an integer (radae_rxe.py get_snrdB_3k_est -> int(), rade_api.c): the device's estimate is within a few 1e-6 dB of the
        # oracle's (float32 sums in another order); when the estimate itself is within 1e-5 dB of an integer the truncation can fall on either side.  Accepted as a tie if snr_int is the
        # ONLY differing output, on calls whose two float estimates agree to 1e-5 dB and lie within 1e-5 dB of that integer boundary (round 6: seen once in 16 seeds, 7.9999986 / 8.0000010)
        stie = False
        if first and set(first) == {"snr_int"} and len(t["snr_int"]) == len(d["snr_int"]):
            calls = [i for i in range(len(t["snr_int"])) if t["snr_int"][i] != d["snr_int"][i]]
            stie = all(abs(float(t["snrdB_3k_est"][i]) - float(d["snrdB_3k_est"][i])) < 1e-5 and abs(float(d["snrdB_3k_est"][i]) - round(float(d["snrdB_3k_est"][i]))) < 1e-5 for i in calls)
            print(f"  (snr_int differs on calls {calls}: estimates {[(float(t['snrdB_3k_est'][i]), float(d['snrdB_3k_est'][i])) for i in calls]})")
        sties += stie
        tties += ttie
        dties += dtie; ties += tie; bad += not (tie or dtie or ttie or stie)
        print(f"{'refine near-tie' if tie else ('detect near-tie' if dtie else ('refine tie between timings' if ttie else ('snr_int truncation tie' if stie else 'MISMATCH')))} case {case}: seed {seed} {chan} Eb/No {eb!r} dB fo {fo!r} Hz valid {nv}/{len(d['features_out'])} max |fmax diff| {dfm:.4f} first differing call per key {first}")
    eng.close()
print(f"{N} cases, {tot_calls} receiver calls, {tot_valid} decoded frames: {bad} mismatching case(s), {ties} with a refine() near-tie resolved the other way, {dties} with a detect_pilots arg-max tie (1 ulp) on an unsynchronised call, {tties} with a refine() tie between two timings (oracle margin < 3e-7, traces equal again within 8 calls), {sties} with an snr_int truncation tie (estimate within 1e-5 dB of an integer)")
if os.environ.get("SWEEP_JSON"):
    import json
    json.dump({"tool": "tools/parity_sweep.py", "seed": int(os.environ.get("SWEEP_SEED", "2026")), "cases": N, "receiver_calls": int(tot_calls), "decoded_modem_frames": int(tot_valid),
               "mismatching_cases": int(bad), "refine_near_tie_cases": int(ties), "detect_argmax_tie_cases": int(dties), "refine_tie_between_timings_cases": int(tties), "snr_int_truncation_tie_cases": int(sties),
               "rule": "per-call discrete outputs equal and features within 1e-4 RMS; a case whose discrete outputs are all equal but whose fmax differs by < 0.05 Hz is a refine() tie if, in addition, the oracle's own arg-max margin (runner-up cell relative to the winner) at the first differing call is below 3e-7 (two 0.1 Hz bins whose float32 magnitudes are within an ulp: summation order decides); a case whose only differing outputs are (tmax, f_ind_max) of unsynchronised calls whose maxima agree to 1e-6 is a detect_pilots arg-max tie (two of the 38,400 float32 cells within one ulp; FFT convolution and direct sums round differently), every later output being equal again; a case whose first differing call is a synchronised one at which the oracle's own refine() arg-max margin is below 3e-7 (two cells of different timing with equal float32 magnitudes), with the same number of decoded frames and traces that are equal again within 8 calls and to the end, is a refine() tie between timings"},
              open(os.environ["SWEEP_JSON"], "w"), indent=1)
