"""Pin the CPU oracle (oracle/rade_oracle.c) against golden vectors captured from the imported
reference (oracle/gen_golden.py).  Tolerances: discrete sync outputs bit-exact; float32 stages
within a few ulp-scaled units; NN outputs within the rounding-noise amplification the reference
itself shows between its stateful and stateless paths (stateful_encoder.py:102 uses 0.01 on loss)."""
import numpy as np
import pytest

RX_CASES = ["awgn", "mpp", "slip_plus", "slip_minus", "foff"]


def rms(a, b):
    return float(np.sqrt(np.mean(np.abs(np.asarray(a, np.complex128) - np.asarray(b, np.complex128)) ** 2)))


def test_constants(oracle, golden):
    c = golden("consts")
    cases = [("w", 30, False, c["w"], 0.0), ("Winv", 9600, True, c["Winv"].ravel(), 2e-9), ("Wfwd", 9600, True, c["Wfwd"].ravel(), 2e-7),
             ("P", 60, True, c["P"], 0.0), ("Pend", 60, True, c["Pend"], 0.0), ("p", 320, True, c["p"], 5e-8), ("pend", 320, True, c["pend"], 5e-8),
             ("p_cp", 384, True, c["p_cp"], 5e-8), ("pend_cp", 384, True, c["pend_cp"], 5e-8), ("eoo", 2304, True, c["eoo_default"], 1e-6),
             ("Pmat", 360, True, c["Pmat"].ravel(), 2e-6), ("bpf_h", 101, False, c["bpf_h"].real, 5e-9),
             ("bpf_phase_vec_exp", 2240, True, c["bpf_phase_vec_exp"], 2e-7), ("acq_p_w", 12800, True, c["acq_p_w"].ravel(), 5e-8),
             ("acq_fcoarse", 40, False, c["acq_fcoarse"], 0.0), ("bpf_alpha", 1, False, c["bpf_alpha"], 0.0), ("pilot_gain", 1, False, c["pilot_gain"], 1e-6)]
    for name, n, cplx, ref, tol in cases:
        got = oracle.get_const(name, n, cplx)
        assert np.abs(got - ref).max() <= tol, name


def test_bpf_golden(oracle, golden):
    """orc_bpf_run against complex_bpf.bpf (dsp.py:63-102) driven call by call with 960 / 800 / 1120-sample calls (oracle/gen_golden_r4.py):
    float32 sums of 101 terms in another order than NumPy's dot."""
    g = golden("bpf")
    bp = oracle.Bpf()
    pos, out = 0, []
    for k in g["sizes"]:
        out.append(bp.run(g["x"][pos:pos + k])); pos += int(k)
    y = np.concatenate(out)
    assert len(y) == len(g["y"]) and set(g["sizes"].tolist()) == {800, 960, 1120}
    assert np.abs(y - g["y"]).max() < 2e-6 * np.abs(g["y"]).max() and rms(y, g["y"]) < 3e-7 * np.abs(g["y"]).max()


def test_txbpf_golden(oracle, oracle_model, golden):
    """radae_tx(..., txbpf_en=True) (radae_txe.py:74-83, :130-132, :141-143; ctest radae_tx_basic): six frames and the end-of-over frame through the Tx
    band-pass filter and the magnitude clip (oracle/gen_golden_r4.py)."""
    g = golden("txbpf")
    tx = oracle.Tx(oracle_model); tx.set_txbpf(True)
    out = np.concatenate([tx.frame(g["features"][12 * k:12 * k + 12].ravel())[0] for k in range(6)])
    assert np.abs(out - g["tx"]).max() < 2e-5 and rms(out, g["tx"]) < 5e-6
    assert np.abs(tx.eoo() - g["eoo"]).max() < 2e-5
    assert np.abs(g["tx"]).max() <= 1.0 + 1e-6 and np.abs(out).max() <= 1.0 + 1e-6          # the clip
    plain = oracle.Tx(oracle_model)
    ref = np.concatenate([plain.frame(g["features"][12 * k:12 * k + 12].ravel())[0] for k in range(6)])
    assert rms(out, ref) > 1e-2                                                            # (the option really changes the signal: 50-sample group delay)


def test_encoder_and_tx(oracle, oracle_model, golden):
    e = golden("enc_tx")
    for u in range(2):
        tx = oracle.Tx(oracle_model)
        zs, txs = [], []
        for k in range(10):
            o, z = tx.frame(e["features"][u, 12 * k:12 * k + 12].ravel())
            zs.append(z); txs.append(o)
        zs = np.array(zs).reshape(30, 80)
        ref_gap = rms(e["z_stateless"][u], e["z"][u])   # the reference's own stateful/stateless gap
        assert rms(zs, e["z"][u]) < 1e-4 and rms(zs, e["z"][u]) < 3 * ref_gap
        assert np.abs(zs - e["z"][u]).max() < 2e-6 * np.abs(e["z"][u]).max() + 1e-5
        assert np.abs(np.array(txs) - e["tx"][u]).max() < 2e-5
        # transmitter alone on the reference's z: float32 rounding only
        t2 = np.array([oracle.ofdm_mod(e["z"][u].reshape(10, 240)[k]) for k in range(10)])
        assert np.abs(t2 - e["tx"][u]).max() < 5e-6
    enc = oracle.Encoder(oracle_model)
    f = e["features"][0]
    for s in range(30):
        feat = np.concatenate([np.concatenate([f[4 * s + i, :20], [-1.0]]) for i in range(4)])
        enc.step(feat)
    for l in range(5):
        assert np.abs(enc.gru_state(l + 1) - e["gru_states_u0"][l]).max() < 2e-5


def test_eoo_bits(oracle, oracle_model, golden):
    c = golden("consts")
    tx = oracle.Tx(oracle_model)
    assert np.abs(tx.eoo() - c["eoo_default"]).max() < 1e-6
    tx.set_eoo_bits(c["eoo_bits_in"])
    assert np.abs(tx.eoo() - c["eoo_with_bits"]).max() < 1e-6


@pytest.mark.parametrize("name", ["mpp", "awgn", "dfdt_pos", "dfdt_neg"])
def test_channel(oracle, oracle_model, golden, name):
    """dfdt_*: df_dt = +-0.5 Hz/s (radae.py:547-552, inference.py:270; ctest radae_rx_dfdt): the per-sample float32 omega summed like
    torch.cumsum, restarted for the EOO frame."""
    g = golden("chan_" + name)
    sigma = float(g["sigma"])
    assert oracle.lib().orc_sigma_from_EbNodB(float(g["EbNodB"])) == pytest.approx(sigma, rel=1e-7)
    rx, fin = oracle.channel(g["tx"], g["G"] if "G" in g else None, g["noise"], sigma, float(g["freq_offset"]), float(g["df_dt"]))
    assert np.abs(rx - g["rx"]).max() < 2e-6
    assert abs(fin - g["final_phase"].ravel()[0]) < 1e-6
    eoo = oracle.channel_eoo(oracle.Tx(oracle_model).eoo(), g["noise_eoo"], sigma, float(g["freq_offset"]), float(g["df_dt"]), fin)
    full = np.concatenate([sigma * g["noise_pre"], rx, eoo, sigma * g["noise_post"]]).astype(np.complex64)
    assert np.abs(full - g["rx_full"]).max() < 5e-6


@pytest.mark.parametrize("name", RX_CASES + ["dfdt", "nounsync"])
def test_rx_trace(oracle, oracle_model, golden, name):
    """dfdt: a drifting offset (ctest radae_rx_dfdt); nounsync: --disable_unsync (radae_rxe.py:277-281) holding sync through a fade
    that the same samples lose it in without the flag."""
    g = golden("rxtrace_" + name)
    du = float(g["disable_unsync"]) if "disable_unsync" in g else 0.0
    d = oracle.run_rx_stream(oracle_model, g["rx_in"], 1, 10.0 if name == "foff" else 0.0, du)
    if du:
        off = oracle.run_rx_stream(oracle_model, g["rx_in"])
        assert not np.array_equal(off["state_after"], g["state_after"])        # the flag is what keeps the fixture in sync
    for k in ["state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]:
        assert np.array_equal(d[k], g[k]), k     # the discrete "indices": bit-exact
    assert np.abs(d["fmax"] - g["fmax"]).max() < 1e-9
    for k in ["Dthresh", "Dtmax12", "Dtmax12_eoo", "snrdB_3k_est"]:
        assert np.abs(d[k] - g[k]).max() < 2e-5, k
    assert d["z_hat"].shape == g["z_hat"].shape
    assert rms(d["z_hat"], g["z_hat"]) < 1e-4
    assert rms(d["features_out"], g["features_out"]) < 1e-5 and np.abs(d["features_out"] - g["features_out"]).max() < 1e-4
    if g["eoo_out"].size:
        assert np.abs(d["eoo_out"] - g["eoo_out"]).max() < 1e-4
        # the aux/EOO bit decisions are the discrete part
        assert np.array_equal(d["eoo_out"] > 0, g["eoo_out"] > 0)

def edge_case_input(g):
    """The round-6 streaming fixtures carry what the reference's ctests pipe into the receiver: int16 samples, converted as the ctest converts them
    (`int16tof32.py --zeropad` for the mono slip + drops file, `int16tof32.py` for the two-channel 8001 Hz file), or complex64 samples."""
    from radae_amd import wire
    if "rx_in" in g.files:
        return g["rx_in"]
    i16 = g["rx_i16"]
    return np.frombuffer(wire.int16_to_f32(np.ascontiguousarray(i16).tobytes(), zeropad=(i16.ndim == 1)), np.complex64)


def check_edge_trace(d, g, zhat_tol=1e-4):
    for k in ["state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]:
        assert np.array_equal(d[k], g[k]), (k, int(np.argmax(d[k][:len(g[k])] != g[k][:len(d[k])])) if len(d[k]) == len(g[k]) else (len(d[k]), len(g[k])))
    assert np.abs(d["fmax"] - g["fmax"]).max() < 1e-9
    for k in ["Dthresh", "Dtmax12", "Dtmax12_eoo", "snrdB_3k_est"]:
        assert np.abs(d[k] - g[k]).max() < 3e-5 * max(1.0, np.abs(g[k]).max()), k
    rows = g["rows_kept"] if "rows_kept" in g.files else np.arange(len(g["features_out"]))
    assert len(d["features_out"]) == (int(g["n_valid_total"]) if "n_valid_total" in g.files else len(g["features_out"]))
    zs = np.abs(g["z_hat"]).max()
    assert rms(d["z_hat"][rows], g["z_hat"]) < zhat_tol * max(1.0, zs)
    fo = d["features_out"][rows]
    assert rms(fo, g["features_out"]) < 1e-5 and np.abs(fo - g["features_out"]).max() < 1e-4
    if g["eoo_out"].size:
        assert d["eoo_out"].shape == g["eoo_out"].shape and np.array_equal(d["eoo_out"] > 0, g["eoo_out"] > 0)


@pytest.mark.parametrize("name", ["slipdrops", "dfs8001", "eoo_mpp"])
def test_rx_trace_edge_cases(oracle, oracle_model, golden, name):
    """The reference's remaining streaming ctests as traces (oracle/gen_golden_r6.py): radae_rx_slip_plus_drops (CMakeLists.txt:397-407; 61 s at 8020 Hz, three
    drop-outs, re-sync after each loss, final state sync), radae_rx_dfs (:374-382; 8001 Hz), radae_eoo_data_mpp (:595-607; EOO data bits through MPP)."""
    g = golden("rxtrace_" + name)
    d = oracle.run_rx_stream(oracle_model, edge_case_input(g))
    check_edge_trace(d, g)
    if name == "slipdrops":
        st = d["state_after"]
        assert st[-1] == 2 and np.sum((st[1:] == 2) & (st[:-1] != 2)) >= 2 and (d["nin_after"] == 1120).sum() >= 5
    if name == "dfs8001":
        assert (d["nin_after"] == 1120).any()
    if name == "eoo_mpp":
        ber = np.array([np.mean(e * g["tx_bits"] < 0) for e in d["eoo_out"]])
        assert np.allclose(ber, g["eoo_ber"]) and ber.min() < 0.05 and len(ber) == 5      # the ctest's pass rule: one over below 5 %


def test_bypass_fixture(oracle, oracle_model, golden):
    """bypass.npz (radae_rxe.py --bypass_dec, radae_txe.py --bypass_enc): the oracle's per-call latents are what the reference wrote out, and its modulator on
    supplied latents reproduces the reference's transmit frames."""
    g = golden("bypass")
    d = oracle.run_rx_stream(oracle_model, g["rx_in"])
    assert np.array_equal(d["ret"], g["ret"]) and np.array_equal(d["state_after"], g["state_after"])
    assert rms(d["z_hat"], g["z_hat_out"]) < 1e-4
    tx = np.array([oracle.ofdm_mod(g["z_in"][k]) for k in range(6)])
    assert np.abs(tx - g["tx"]).max() < 5e-6


def test_decoder_and_loss(oracle, oracle_model, golden):
    g = golden("dec_loss")
    dec = oracle.Decoder(oracle_model)
    out = np.array([dec.step(g["z_hat"][k]) for k in range(30)]).reshape(120, 21)
    assert rms(out, g["features"]) < 1e-5 and np.abs(out - g["features"]).max() < 1e-4
    for l in range(5):
        assert np.abs(dec.gru_state(l + 1) - g["gru_states"][l]).max() < 2e-5
    assert oracle.distortion_loss(g["la"], g["lb"], 20) == pytest.approx(float(g["loss20"]), rel=2e-6)
    assert oracle.distortion_loss(g["la21"], g["lb21"], 21) == pytest.approx(float(g["loss21"]), rel=2e-6)


def test_loopback_loss_alignment(oracle, oracle_model, golden):
    """loss.py-style aligned loss between features_in and the oracle receiver's output equals the
    one computed from the reference's own output (delta < 1e-4)."""
    g = golden("rxtrace_awgn")
    d = oracle.run_rx_stream(oracle_model, g["rx_in"])
    fi = g["features_in"]
    l_ref, s_ref = oracle.find_loss(fi, g["features_out"].reshape(-1, 36))
    l_orc, s_orc = oracle.find_loss(fi, d["features_out"].reshape(-1, 36))
    assert s_ref == s_orc and abs(l_ref - l_orc) < 1e-4


def test_model05_rate_rs_config1(oracle, golden):
    """BASELINE config 1: model05 through the rate-Rs channel (inference.py plumbing), oracle vs the reference."""
    import os
    g = golden("model05")
    m = oracle.Model(os.path.join(os.path.dirname(oracle.BLOB), "model05.bin"))
    enc = oracle.Encoder(m)
    f = g["features"].reshape(-1, 80)
    z = np.array([enc.step(f[t], bottleneck=1) for t in range(len(f))])
    assert rms(z, g["z"]) < 2e-5 and np.abs(z).max() <= 1.0
    for tag in ("awgn", "mp"):
        zh = oracle.channel_rs(g["z"], g[tag + "_H"], g[tag + "_noise"], float(g[tag + "_sigma"])).reshape(-1, 80)
        assert np.abs(zh - g[tag + "_z_hat"]).max() < 1e-6
        dec = oracle.Decoder(m)
        fh = np.array([dec.step(g[tag + "_z_hat"][t]) for t in range(len(f))]).reshape(-1, 20)
        assert rms(fh, g[tag + "_features_hat"]) < 2e-5
        assert oracle.distortion_loss(g["features"], fh, 20) == pytest.approx(float(g[tag + "_loss"]), abs=1e-4)


def test_bbfm_config5(oracle, golden):
    """BASELINE config 5: BBFM.forward (bbfm.py:157-197) with the seeded random-weight blob, oracle vs reference."""
    import os
    g = golden("bbfm")
    m = oracle.Model(os.path.join(os.path.dirname(oracle.BLOB), "bbfm_random_seed20240501.bin"))
    enc = oracle.Encoder(m)
    f = g["features"].reshape(-1, 80)
    z = np.array([enc.step(f[t], bottleneck=1) for t in range(len(f))])
    assert rms(z, g["z"]) < 1e-5
    Gfm = float(g["Gfm"])
    for tag in ("awgn", "ray"):
        zh = oracle.channel_bbfm(g["z"], g[tag + "_H"], g[tag + "_noise"], float(g[tag + "_CNRdB"]), Gfm).reshape(-1, 80)
        assert np.abs(zh - g[tag + "_z_hat"]).max() < 2e-6 and np.abs(zh).max() <= 1.0
        dec = oracle.Decoder(m)
        fh = np.array([dec.step(g[tag + "_z_hat"][t]) for t in range(len(f))]).reshape(-1, 20)
        assert rms(fh, g[tag + "_features_hat"]) < 1e-5
