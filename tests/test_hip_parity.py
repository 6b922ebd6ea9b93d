"""GPU parity tests (-m gpu): the HIP engine, called through the C ABI (rade_batch.h / rade_api.h),
against (1) golden vectors captured from the imported reference and (2) the CPU oracle on fresh inputs.

Bars (BASELINE.json north_star): discrete sync outputs bit-exact; float32 features within 1e-4 RMS."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

INT_KEYS = ["state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]
RX_CASES = ["awgn", "mpp", "slip_plus", "slip_minus", "foff"]


def rms(a, b):
    return float(np.sqrt(np.mean(np.abs(np.asarray(a, np.complex128) - np.asarray(b, np.complex128)) ** 2)))


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def Engine():
    from radae_amd.engine import BatchEngine
    return BatchEngine


def test_encoder_tx_golden(Engine, torch_dev, golden):
    import torch
    e = golden("enc_tx")
    f = torch.tensor(e["features"], device=torch_dev)
    eng = Engine(2, max_tx_mf=10)
    iq, z = eng.tx(f, want_z=True)
    zr, txr = e["z"], e["tx"]
    assert rms(z.cpu().numpy(), zr) < 1e-4
    assert np.abs(z.cpu().numpy() - zr).max() < 2e-6 * np.abs(zr).max() + 1e-5
    assert np.abs(iq.cpu().numpy().reshape(2, 10, 960) - txr).max() < 2e-5
    # streaming == whole utterance, bit for bit (stateful == stateless, ctest stateful_encoder)
    eng.tx_reset()
    parts = [eng.tx(f[:, 12 * k:12 * k + 12].contiguous(), want_z=True) for k in range(10)]
    assert torch.equal(torch.cat([p[0] for p in parts], 1), iq)
    assert torch.equal(torch.cat([p[1] for p in parts], 1), z)
    eng.tx_reset()
    parts = [eng.tx(f[:, 12 * a:12 * b].contiguous()) for a, b in ((0, 3), (3, 4), (4, 10))]
    assert torch.equal(torch.cat(parts, 1), iq)
    eng.close()


def test_encoder_large_batch_split_f16_gemm(Engine, torch_dev, oracle, oracle_model):
    """More than 16 k GEMM rows selects the split-binary16 matrix-core kernels (k_gemm16, 22-bit operands, f32
    accumulation): the latents match the oracle's float32 encoder to ~6e-6 of full scale (f32 kernels: ~1e-6), and the
    transmit samples to 5e-6 RMS per stream (single samples up to 1.1e-4: see the end of the test)."""
    import torch
    from radae_amd.channel_tools import synth_features
    B, n_mf = 72, 84                                          # 72 x 252 = 18144 rows
    feats = np.stack([synth_features(700 + b, 12 * n_mf) for b in range(B)])
    eng = Engine(B, max_tx_mf=n_mf)
    iq, z = eng.tx(torch.tensor(feats, device=torch_dev), want_z=True)
    z = z.cpu().numpy(); iq = iq.cpu().numpy()
    eng.close()
    small = Engine(1, max_tx_mf=n_mf)
    worst_max = worst_rms = 0.0
    for b in range(B):                                        # every stream against the oracle (all time tiles, both ends of the batch); three of them against the one-stream engine
        tx = oracle.Tx(oracle_model)
        ref = [tx.frame(feats[b, 12 * k:12 * k + 12].ravel()) for k in range(n_mf)]
        zr = np.concatenate([r[1] for r in ref]).reshape(-1, 80); sig = np.concatenate([r[0] for r in ref])
        dz = np.abs(z[b].reshape(-1, 80) - zr).max() / np.abs(zr).max()
        assert dz < 2e-5, dz                                  # latents are O(100) (bottleneck 3 is linear); 22-bit operands: ~6e-6 measured
        worst_max = max(worst_max, float(np.abs(iq[b] - sig).max())); worst_rms = max(worst_rms, rms(iq[b], sig))
        if b not in (0, 35, 71): continue
        small.tx_reset()
        zs = small.tx(torch.tensor(feats[b][None], device=torch_dev), want_z=True)[1].cpu().numpy()[0]
        dz = np.abs(zs - z[b]).max() / np.abs(zr).max()
        assert dz < 2e-5, dz
    small.close()
    # all 72 streams x 80,640 samples (round 5; rounds 2-4 looked at three streams and a 5e-5 bar held for those): the recurrences carry the 22-bit operand roundings
    # along 252 steps, the worst single sample is 1.1e-4 off, the worst stream 1.6e-6 RMS (the float32-row and the fragment kernels give the same figures: same bits)
    assert worst_max < 3e-4 and worst_rms < 5e-6, (worst_max, worst_rms)


def test_encoder_fragment_layout_equals_row_layout(Engine, torch_dev, monkeypatch):
    """Calls with more than 16 k GEMM rows keep the encoder's concat buffer as matrix-core operand fragments (rade_enc.hip: two binary16
    planes per activation, written once by the producing layer); $RADE_ENC_ROWS keeps the float32-row kernels (k_gemm16p) for every size.
    With the conv taps summed one after the other ($RADE_ENCF_SEQ_TAPS: the float32-row kernels' order) the products and their order are the
    same: latents and transmit samples bit-identical over consecutive long calls (conv history through the history tile).  The shipped
    order alternates the taps per k-block (tap 1 re-reads tap 0's cache lines while they are hot), the fused launches ($RADE_ENCF_FUSED)
    do too, and a short call (float32 rows, k_gemm_splitk) between two long ones takes the history from the 22-bit planes instead of the
    float32 rows: equal to float32 roundings, which the recurrences carry along (759 steps here)."""
    import torch
    from radae_amd.channel_tools import synth_features
    B, n_mf = 72, 84                                          # 72 x 252 = 18144 rows
    feats = torch.tensor(np.stack([synth_features(900 + b, 12 * (3 * n_mf + 3)) for b in range(B)]), device=torch_dev)
    cuts = ((0, n_mf), (n_mf, 2 * n_mf), (2 * n_mf, 2 * n_mf + 3), (2 * n_mf + 3, 3 * n_mf + 3))
    def run():
        eng = Engine(B, max_tx_mf=n_mf)
        out = [eng.tx(feats[:, 12 * a:12 * b].contiguous(), want_z=True) for a, b in cuts]
        eng.close()
        return out
    shipped = run()
    monkeypatch.setenv("RADE_ENCF_FUSED", "1")
    fused = run()
    monkeypatch.delenv("RADE_ENCF_FUSED")
    monkeypatch.setenv("RADE_ENCF_SEQ_TAPS", "1")
    seq = run()
    monkeypatch.setenv("RADE_ENCF_NO_PAIR", "1")
    seq_nopair = run()
    monkeypatch.setenv("RADE_ENC_ROWS", "1")
    rows = run()
    for k in (0, 1):
        assert torch.equal(seq[k][1], rows[k][1]) and torch.equal(seq[k][0], rows[k][0])
        assert torch.equal(seq_nopair[k][1], rows[k][1])
    assert not torch.equal(fused[0][1], rows[0][1]) and not torch.equal(shipped[0][1], rows[0][1])     # (the switches did select other kernels)
    for other in (shipped, fused, seq):
        for (iq_f, z_f), (iq_r, z_r) in zip(other, rows):
            assert (z_f - z_r).abs().max().item() < 2e-5 * z_r.abs().max().item()
            assert (iq_f - iq_r).abs().max().item() < 2e-4


def test_encoder_fragment_layout_short_calls_many_streams(Engine, torch_dev, monkeypatch):
    """The same equality where a call is ONE modem frame (3 encoder steps: one partly filled time tile, both conv taps' earlier rows in the
    history tile) of enough streams for the batched kernels: five consecutive calls, bit-identical to the float32-row kernels."""
    import torch
    from radae_amd.channel_tools import synth_features
    B = 5632                                                  # x 3 steps = 16896 rows
    base = np.stack([synth_features(40 + b, 12 * 5) for b in range(64)])
    feats = torch.tensor(np.tile(base, (B // 64, 1, 1)) * (1.0 + 0.001 * (np.arange(B) // 64))[:, None, None].astype(np.float32), device=torch_dev)
    def run():
        eng = Engine(B, max_tx_mf=1)
        out = [eng.tx(feats[:, 12 * k:12 * k + 12].contiguous(), want_z=True) for k in range(5)]
        eng.close()
        return out
    monkeypatch.setenv("RADE_ENCF_SEQ_TAPS", "1")            # (the float32-row kernels' summation order: see the test above)
    frag = run()
    monkeypatch.setenv("RADE_ENC_ROWS", "1")
    rows = run()
    for (iq_f, z_f), (iq_r, z_r) in zip(frag, rows):
        assert torch.equal(z_f, z_r) and torch.equal(iq_f, iq_r)
    assert not torch.equal(frag[4][1][0], frag[4][1][64])     # (the streams are not copies of each other)


def test_encoder_fragment_layout_ragged_call_lengths(Engine, torch_dev, monkeypatch):
    """Calls of 150, 252 and 138 steps on an engine sized for 252 (partly filled last tiles, tiles left untouched, the history taken from
    a different tile each time): bit-identical to the float32-row kernels with the conv taps in their order."""
    import torch
    from radae_amd.channel_tools import synth_features
    B, cap = 120, 84
    lens = (50, 84, 46)                                       # x 3 steps x 120 streams = 18000 / 30240 / 16560 rows: all batched
    feats = torch.tensor(np.stack([synth_features(300 + b, 12 * sum(lens)) for b in range(B)]), device=torch_dev)
    def run():
        eng = Engine(B, max_tx_mf=cap)
        out, a = [], 0
        for n in lens:
            out.append(eng.tx(feats[:, 12 * a:12 * (a + n)].contiguous(), want_z=True)); a += n
        eng.close()
        return out
    monkeypatch.setenv("RADE_ENCF_SEQ_TAPS", "1")
    frag = run()
    monkeypatch.setenv("RADE_ENC_ROWS", "1")
    rows = run()
    for (iq_f, z_f), (iq_r, z_r) in zip(frag, rows):
        assert torch.equal(z_f, z_r) and torch.equal(iq_f, iq_r)


def test_eoo_frames(Engine, golden):
    c = golden("consts")
    eng = Engine(3, max_tx_mf=1)
    assert np.abs(eng.tx_eoo().cpu().numpy()[2] - c["eoo_default"]).max() < 1e-6
    bits = np.stack([c["eoo_bits_in"], -c["eoo_bits_in"], c["eoo_bits_in"]])
    eng.set_eoo_bits(bits)
    out = eng.tx_eoo().cpu().numpy()
    assert np.abs(out[0] - c["eoo_with_bits"]).max() < 1e-6 and np.abs(out[2] - c["eoo_with_bits"]).max() < 1e-6
    assert np.abs(out[1] - c["eoo_with_bits"]).max() > 0.1
    eng.set_eoo_bits(None)
    assert np.abs(eng.tx_eoo().cpu().numpy()[0] - c["eoo_default"]).max() < 1e-6
    eng.close()


@pytest.mark.parametrize("name", ["mpp", "awgn"])
def test_channel_golden(Engine, torch_dev, golden, name):
    import torch
    g = golden("chan_" + name)
    eng = Engine(1, max_tx_mf=1)
    sigma = float(g["sigma"])
    from radae_amd.engine import sigma_from_EbNodB
    assert sigma_from_EbNodB(float(g["EbNodB"])) == pytest.approx(sigma, rel=1e-6)
    noise = np.concatenate([g["noise_pre"].astype(np.complex64), g["noise"], g["noise_eoo"], g["noise_post"].astype(np.complex64)])
    rx = eng.channel(torch.tensor(g["tx"][None], device=torch_dev), sigma, float(g["freq_offset"]), len(g["noise_pre"]), len(g["noise_post"]), True,
                     G=torch.tensor(g["G"][None], device=torch_dev), noise=torch.tensor(noise[None], device=torch_dev))
    assert np.abs(rx.cpu().numpy()[0] - g["rx_full"]).max() < 1e-5
    eng.close()


@pytest.mark.parametrize("name", ["dfdt_pos", "dfdt_neg"])
def test_channel_df_dt_golden(Engine, torch_dev, golden, name):
    """Frequency drift (radae.py:547-552, inference.py:270; ctest radae_rx_dfdt) against RADAE.forward's output.  The kernel
    evaluates the phase sum in closed form (a thread cannot wait for 90 k predecessors); it differs from the reference's sum of
    per-sample float32 omegas by <= 2e-7 rad before both are rounded to float32 -- at ~800 rad one float32 step is 6e-5 rad, so a
    few samples land on the neighbouring float32 phase: RMS stays far inside 1e-5, single samples within 6e-5 x |sample|."""
    import torch
    g = golden("chan_" + name)
    eng = Engine(1, max_tx_mf=1)
    sigma = float(g["sigma"])
    noise = np.concatenate([g["noise_pre"].astype(np.complex64), g["noise"], g["noise_eoo"], g["noise_post"].astype(np.complex64)])
    rx = eng.channel(torch.tensor(g["tx"][None], device=torch_dev), sigma, float(g["freq_offset"]), len(g["noise_pre"]), len(g["noise_post"]), True,
                     noise=torch.tensor(noise[None], device=torch_dev), df_dt=float(g["df_dt"]))
    d = rx.cpu().numpy()[0] - g["rx_full"]
    assert rms(d, 0 * d) < 1e-5 and np.abs(d).max() < 2e-4
    assert np.mean(np.abs(d) > 1e-5) < 0.01
    eng.close()


@pytest.mark.parametrize("name", RX_CASES + ["dfdt", "nounsync"])
def test_rx_trace_golden(Engine, torch_dev, golden, name, monkeypatch):
    import torch
    g = golden("rxtrace_" + name)
    du = float(g["disable_unsync"]) if "disable_unsync" in g else 0.0       # radae_rxe.py --disable_unsync (ctests radae_rx_mpp / _mpg)
    eng = Engine(1, max_tx_mf=1, rx_trace_calls=64, flags=(4 if name == "foff" else 0), disable_unsync=du)
    feats, st, eoo = eng.rx(torch.tensor(g["rx_in"][None], device=torch_dev))
    d = eng.rx_trace(0)
    for k in INT_KEYS:
        assert np.array_equal(d[k], g[k]), k                       # discrete outputs: bit-exact
    assert np.abs(d["fmax"] - g["fmax"]).max() < 1e-9
    for k in ["Dthresh", "Dtmax12", "Dtmax12_eoo", "snrdB_3k_est"]:
        assert np.abs(d[k] - g[k]).max() < 3e-5, k
    assert rms(d["z_hat"], g["z_hat"]) < 1e-4
    nv = st[0].n_valid
    assert nv == len(g["features_out"])
    fo = feats.cpu().numpy()[0, :nv]
    assert rms(fo, g["features_out"]) < 1e-5 and np.abs(fo - g["features_out"]).max() < 1e-4     # north star: 1e-4 RMS
    if g["eoo_out"].size:
        assert st[0].has_eoo
        assert np.abs(eoo.cpu().numpy()[0] - g["eoo_out"][-1]).max() < 1e-4
        assert np.array_equal(eoo.cpu().numpy()[0] > 0, g["eoo_out"][-1] > 0)                   # EOO bit decisions
    eng.close()


@pytest.mark.parametrize("name", ["slipdrops", "dfs8001", "eoo_mpp"])
def test_rx_trace_edge_cases(Engine, torch_dev, golden, name):
    """The reference's remaining streaming ctests as golden traces (oracle/gen_golden_r6.py), whole trace bit-exact in one receiver launch:
    slipdrops = radae_rx_slip_plus_drops (CMakeLists.txt:397-407): 61 s of a real-valued int16 signal at 8020 Hz with three drop-outs; sync is lost and regained, 1120-sample
    calls all along, the last state is sync.  dfs8001 = radae_rx_dfs (:374-382).  eoo_mpp = radae_eoo_data_mpp (:595-607): the data bits of five end-of-over
    frames through MPP at the ctest's SNR, decisions equal to the reference's, one over below 5 % BER."""
    import torch
    from test_oracle_golden import check_edge_trace, edge_case_input
    g = golden("rxtrace_" + name)
    x = edge_case_input(g)
    eng = Engine(1, max_tx_mf=1, rx_trace_calls=len(g["ret"]) + 8)
    feats, st, eoo = eng.rx(torch.tensor(x[None], device=torch_dev))
    d = eng.rx_trace(0)
    nv = st[0].n_valid
    d["features_out"] = feats.cpu().numpy()[0, :nv]
    assert st[0].n_calls == len(g["ret"]) and st[0].state == g["state_after"][-1]
    check_edge_trace(d, g)
    if name == "slipdrops":
        assert st[0].state == 2 and st[0].sync == 1                      # the ctest's `grep 'state: sync'`
    if name == "eoo_mpp":
        ber = np.array([np.mean(e * g["tx_bits"] < 0) for e in d["eoo_out"]])
        assert np.allclose(ber, g["eoo_ber"]) and ber.min() < 0.05 and np.abs(d["eoo_out"] - g["eoo_out"]).max() < 1e-4 * np.abs(g["eoo_out"]).max() + 1e-4
        assert np.array_equal(eoo.cpu().numpy()[0] > 0, g["eoo_out"][-1] > 0)
    eng.close()


def test_bypass_dec_and_bypass_enc(Engine, torch_dev, golden):
    """`radae_rxe.py --bypass_dec` (radae_rxe.py:300-302, :315) = an engine opened with RADE_BATCH_BYPASS_DEC: 240 latents per valid modem frame out, no decoder, no UW
    accounting; `radae_txe.py --bypass_enc` (radae_txe.py:124-126) = rade_batch_tx_latents.  Fixture: oracle/gen_golden_r6.py (the awgn trace's samples)."""
    import torch
    from radae_amd.engine import BYPASS_DEC
    g = golden("bypass")
    eng = Engine(1, max_tx_mf=6, rx_trace_calls=64, flags=BYPASS_DEC)
    rows, st, eoo = eng.rx(torch.tensor(g["rx_in"][None], device=torch_dev))
    assert rows.shape[2] == 240
    d = eng.rx_trace(0)
    assert np.array_equal(d["ret"], g["ret"]) and np.array_equal(d["state_after"], g["state_after"]) and np.array_equal(d["uw_errors"], g["uw_errors"]) and not d["uw_errors"].any()
    nv = st[0].n_valid
    assert nv == len(g["z_hat_out"])
    z = rows.cpu().numpy()[0, :nv]
    assert rms(z, g["z_hat_out"]) < 1e-4 and np.abs(z - g["z_hat_out"]).max() < 1e-3 and rms(z, 0 * z) > 0.3
    assert st[0].has_eoo and np.array_equal(eoo.cpu().numpy()[0] > 0, g["eoo_out"][-1] > 0)
    # rows beyond n_valid untouched; a short buffer pauses the stream exactly like the 432-float mode
    assert not rows.cpu().numpy()[0, nv:].any()
    eng.rx_reset()
    small = torch.zeros((1, 3, 240), dtype=torch.float32, device=torch_dev)
    _, st2, _ = eng.rx(torch.tensor(g["rx_in"][None], device=torch_dev), features_out=small)
    assert st2[0].n_valid == 3 and np.array_equal(small.cpu().numpy()[0], z[:3]) and st2[0].consumed < g["rx_in"].size
    # latents in -> transmit frames
    tx = eng.tx_latents(torch.tensor(g["z_in"].reshape(1, 18, 80), device=torch_dev)).cpu().numpy().reshape(6, 960)
    assert np.abs(tx - g["tx"]).max() < 2e-5 and rms(tx, g["tx"]) < 5e-6
    eng.close()
    # the reference classes' own call pattern (radae_rxe.py:349-356, radae_txe.py:165-176) through the Python mirrors of the bypass modes
    from radae_amd import api
    rxb = api.radae_rx_bypass_dec()
    fo = np.zeros(rxb.get_n_floats_out(), np.float32); assert fo.size == 240
    x, pos, rets, zs = g["rx_in"], 0, [], []
    while pos + rxb.get_nin() <= len(x):
        nin = rxb.get_nin(); ret = rxb.do_radae_rx(x[pos:pos + nin], fo); pos += nin; rets.append(ret)
        if ret & 1: zs.append(fo.copy())
        if ret & 2: assert np.array_equal(fo[:180] > 0, g["eoo_out"][-1] > 0)
    assert np.array_equal(np.array(rets), g["ret"]) and rms(np.array(zs), g["z_hat_out"]) < 1e-4
    rxb.eng.close()
    txb = api.radae_tx_bypass_enc()
    out = np.zeros(960, np.complex64)
    for k in range(6):
        txb.do_radae_tx(g["z_in"][k], out)
        assert np.abs(out - g["tx"][k]).max() < 2e-5
    txb.eng.close()
    # the same latents through an engine with the Tx band-pass filter differ (the filter is applied), and the decoder-side default is untouched
    plain = Engine(1, max_tx_mf=6, rx_trace_calls=64)
    f, stp, _ = plain.rx(torch.tensor(g["rx_in"][None], device=torch_dev))
    assert f.shape[2] == 432 and stp[0].n_valid == nv
    plain.close()


@pytest.mark.parametrize("name", ["slip_plus", "slip_minus", "mpp"])
def test_bpf_prepass_matches_oracle_filter(Engine, torch_dev, golden, oracle, name):
    """The band-pass filter runs ahead of the receiver kernel for a whole invocation (k_rx_bpf); a stream whose nin changes inside an invocation
    (timing slips) filters the rest itself (rx2_bpf_own).  Either way the receiver must have read what complex_bpf.bpf (dsp.py:63-102) produces
    for the stream's actual sequence of calls: compared with the oracle's filter driven with the traced call sizes (float32 rounding of a
    101-term sum), and bit for bit with the same stream fed one call per invocation (where every call is on the pre-pass's grid)."""
    import torch
    g = golden("rxtrace_" + name)
    x = g["rx_in"]
    eng = Engine(1, max_tx_mf=1, rx_trace_calls=128)
    eng.rx(torch.tensor(x[None], device=torch_dev))
    sizes = eng.rx_trace(0)["nin_before"]
    assert np.array_equal(sizes, g["nin_before"])
    if name != "mpp":
        assert (sizes != 960).any()                     # the case really leaves the grid
    n = int(sizes.sum())
    y = eng.rx_filtered(0, n)
    bp = oracle.Bpf()
    pos, ref = 0, []
    for k in sizes:
        ref.append(bp.run(x[pos:pos + k])); pos += int(k)
    ref = np.concatenate(ref)
    assert np.abs(y - ref).max() < 2e-6 * np.abs(ref).max()
    eng.rx_reset()
    pos, outs = 0, []
    for k in sizes:
        _, s, _ = eng.rx(torch.tensor(x[None, pos:pos + k], device=torch_dev), max_calls=1)
        assert s[0].consumed == k
        outs.append(eng.rx_filtered(0, int(k))); pos += int(k)
    assert np.array_equal(np.concatenate(outs), y)
    eng.close()


def test_bpf_prepass_golden(Engine, torch_dev, golden):
    """The pre-pass against the reference's own filter output (tests/golden/bpf.npz): the 960-sample calls a receiver in the search state makes."""
    import torch
    g = golden("bpf")
    n = 3 * 960
    assert list(g["sizes"][:3]) == [960, 960, 960]
    eng = Engine(1, max_tx_mf=1)
    _, s, _ = eng.rx(torch.tensor(g["x"][None, :n], device=torch_dev))
    assert s[0].consumed == n
    y = eng.rx_filtered(0, n)
    assert np.abs(y - g["y"][:n]).max() < 2e-6 * np.abs(g["y"]).max()
    eng.close()


def test_tx_bpf_and_clip_golden(Engine, torch_dev, golden, oracle, oracle_model):
    """RADE_BATCH_TX_BPF = radae_tx(..., txbpf_en=True) (radae_txe.py:74-83, :130-132, :141-143; ctest radae_tx_basic): every frame and the end-of-over frame
    through the Tx band-pass filter and the magnitude clip.  Against the reference's own output (tests/golden/txbpf.npz) and the oracle; six frames in one call
    == six calls of one frame, bit for bit (the filter state is carried); two streams with different inputs stay independent."""
    import torch
    g = golden("txbpf")
    n_mf = 6
    f2 = np.stack([g["features"], g["features"][::-1].copy()])        # stream 1: another utterance (the same frames in reverse order)
    eng = Engine(2, max_tx_mf=n_mf, flags=0x400)                      # RADE_BATCH_TX_BPF
    iq = eng.tx(torch.tensor(f2, device=torch_dev)).cpu().numpy()
    eoo = eng.tx_eoo().cpu().numpy()
    assert np.abs(iq[0] - g["tx"]).max() < 2e-5 and rms(iq[0], g["tx"]) < 5e-6
    assert np.abs(eoo[0] - g["eoo"]).max() < 2e-5
    assert np.abs(iq).max() <= 1.0 + 1e-6 and np.abs(eoo).max() <= 1.0 + 1e-6
    tx = oracle.Tx(oracle_model); tx.set_txbpf(True)
    o1 = np.concatenate([tx.frame(f2[1, 12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)])
    assert np.abs(iq[1] - o1).max() < 2e-5 and np.abs(eoo[1] - tx.eoo()).max() < 2e-5
    eng.tx_reset()
    parts = [eng.tx(torch.tensor(f2[:, 12 * k:12 * k + 12].copy(), device=torch_dev)).cpu().numpy() for k in range(n_mf)]
    assert np.array_equal(np.concatenate(parts, axis=1), iq) and np.array_equal(eng.tx_eoo().cpu().numpy(), eoo)
    eng.close()
    plain = Engine(2, max_tx_mf=n_mf)
    assert rms(plain.tx(torch.tensor(f2, device=torch_dev)).cpu().numpy()[0], g["tx"]) > 1e-2      # the option is off by default
    plain.close()


def test_tx_bpf_single_stream_c_abi_and_channel_eoo(Engine, torch_dev, golden):
    """The Tx band-pass filter + clip through the OTHER two doors: (i) the single-stream C ABI -- radae_amd.api.radae_tx(txbpf_en=True) = rade_open(flags |
    RADE_BATCH_TX_BPF), rade_tx() per frame and rade_tx_eoo() against the reference's own radae_tx(txbpf_en=True) output (tests/golden/txbpf.npz; radae_txe.py:74-83,
    :130-132, :141-143); (ii) the channel's with_eoo: the end-of-over frame a receiver sees behind the last frame is the FILTERED one, the filter state carried
    over from the frames (what `radae_tx.py --txbpf | ch` hands to the receiver), both through rade_batch_tx + rade_batch_channel and rade_batch_tx_channel."""
    import torch
    from radae_amd import api
    g = golden("txbpf"); n_mf = 6
    tx = api.radae_tx(txbpf_en=True)
    out = np.zeros(960, np.complex64); eo = np.zeros(1152, np.complex64)
    for k in range(n_mf):
        tx.do_radae_tx(g["features"][12 * k:12 * k + 12].ravel(), out)
        assert np.abs(out - g["tx"][960 * k:960 * k + 960]).max() < 2e-5, k
    tx.do_eoo(eo)
    assert np.abs(eo - g["eoo"]).max() < 2e-5 and np.abs(eo).max() <= 1.0 + 1e-6
    with pytest.raises(ValueError):                   # a handle opened with the filter cannot serve a radae_tx that says it has none (and vice versa): no silent other signal
        api.radae_tx(handle=tx.h, txbpf_en=False)
    tx.h.close()
    f1 = torch.tensor(g["features"][None], device=torch_dev)
    eng = Engine(1, max_tx_mf=n_mf, flags=0x400)
    iq = eng.tx(f1)
    rx = eng.channel(iq, 0.0, 0.0, n_pre=0, n_post=0, with_eoo=True).cpu().numpy()[0]        # no noise, no offset, no multipath: gain 1 is not implied, so compare shapes
    gain = np.vdot(g["tx"], rx[:n_mf * 960]) / np.vdot(g["tx"], g["tx"])                      # the channel's power normalisation of the signal part
    assert abs(gain.imag) < 1e-6 and np.abs(rx[:n_mf * 960] - gain.real * g["tx"]).max() < 3e-5
    assert np.abs(rx[n_mf * 960:n_mf * 960 + 1152] - gain.real * g["eoo"]).max() < 3e-5      # the EOO frame behind it: filtered + clipped, state carried on
    rx_again = eng.channel(iq, 0.0, 0.0, n_pre=0, n_post=0, with_eoo=True).cpu().numpy()[0]  # the channel call only reads the Tx filter state: a second pass gives the same EOO
    assert np.array_equal(rx_again, rx)
    eng.tx_reset()
    rx2 = eng.tx_channel(f1, 0.0, 0.0, n_pre=0, n_post=0, with_eoo=True).cpu().numpy()[0]
    assert np.array_equal(rx2, rx)
    eng.close()


def test_transmit_side_is_bit_reproducible_under_load():
    """Three engines on three HIP streams / host threads (bench.py's pipeline) run the same utterances step after step: every step's transmit samples and
    received samples are bit-identical to the first step's.  (Round 5: with the modulator's IDFT on v_pk_fma_f32 with operand modifiers, one 16-sample block
    in ~15,000 frames differed from run to run under exactly this load -- ~40 blocks in this test's 64 x 42 x 15 frames; rade_devutil.h: idft_term.)"""
    import json, subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "tx_determinism.py"), "6", "64", "504"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["frames_checked"] == 3 * 5 * 64 * 42 and d["mismatching_blocks"] == 0, d["detail"][:4]


def test_blocking_wait_equals_spinning():
    """The host path an 8-GPU job takes under a 16-core quota (8 ranks x 3 engines > CPUs): rade_batch_rx sleeps on a hipEventBlockingSync event instead of
    spinning (rade_engine.c: sync_blocking_now).  Three engines on three streams / host threads, two steps each: RADE_SYNC=block returns bit for bit what
    RADE_SYNC=spin returns, the waits are counted as blocking, and RADE_SYNC_PEERS=64 makes the automatic policy choose blocking by itself."""
    import json, subprocess, sys
    def run(**env):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "sync_mode_check.py")], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    spin, block, auto = run(RADE_SYNC="spin"), run(RADE_SYNC="block"), run(RADE_SYNC_PEERS="64")
    assert spin["rx_waits_blocking"] == 0 and spin["rx_waits_spinning"] > 0
    assert block["rx_waits_blocking"] > 0 and block["rx_waits_spinning"] == 0
    assert auto["rx_waits_blocking"] > 0 and auto["rx_waits_spinning"] == 0
    assert spin["sha256"] == block["sha256"] == auto["sha256"] and spin["decoded"] > 0


def test_rx_call_chunking_is_invariant(Engine, torch_dev, golden, monkeypatch):
    """One do_radae_rx call per invocation (the rade_rx() usage) == the whole stream at once."""
    import torch
    g = golden("rxtrace_slip_plus")
    x = torch.tensor(g["rx_in"][None], device=torch_dev)
    eng = Engine(1, max_tx_mf=1, rx_trace_calls=64)
    fa, sa, _ = eng.rx(x)
    ta = eng.rx_trace(0)
    eng.rx_reset()
    pos, outs = 0, []
    nin = 960
    while pos + nin <= x.shape[1]:
        f, s, _ = eng.rx(x[:, pos:pos + nin].contiguous(), max_calls=1)
        assert s[0].n_calls == 1 and s[0].consumed == nin
        if s[0].n_valid:
            outs.append(f[0, 0].cpu().numpy())
        pos += nin; nin = s[0].nin
    tb = eng.rx_trace(0)
    for k in INT_KEYS:
        assert np.array_equal(ta[k], tb[k]), k
    assert np.array_equal(np.array(outs), fa.cpu().numpy()[0, :sa[0].n_valid])
    eng.close()


def test_uw_failures_and_launch_granularity(Engine, torch_dev, oracle, oracle_model, monkeypatch):
    """The decoder stage runs inside the receiver kernel right before each unique-word decision
    (radae_rxe.py:220-224).  A RADE_FOFF_TEST frequency error and low SNR make windows fail.  The result must not
    depend on how the work is cut: one call per launch (RADE_ROUND_CALLS=1), a 3-row decoder buffer (the decoder runs
    after every frame) and the default (whole utterance per launch, decode at UW checks) agree bit for bit with each
    other on everything, and with the oracle on the trace."""
    import torch
    from radae_amd.engine import sigma_from_EbNodB
    n_mf = 40
    streams = []
    for seed, eb, fo in [(31, 20.0, 5.0), (32, -2.0, -20.0), (33, 1.0, 12.0)]:
        feats, G, n_pre, noise = _make_stream(seed, n_mf, eb, fo, "mpp")
        streams.append((feats, G, n_pre, noise, sigma_from_EbNodB(eb), fo))
    res = {}
    for mode in ("1", "rows3", None):
        monkeypatch.delenv("RADE_ROUND_CALLS", raising=False); monkeypatch.delenv("RADE_DEC_ROWS", raising=False)
        if mode == "1": monkeypatch.setenv("RADE_ROUND_CALLS", "1")
        if mode == "rows3": monkeypatch.setenv("RADE_DEC_ROWS", "3"); monkeypatch.setenv("RADE_ROUND_CALLS", "7")
        out = []
        for i, (feats, G, n_pre, noise, sigma, fo) in enumerate(streams):
            eng = Engine(1, max_tx_mf=n_mf, rx_trace_calls=64, flags=(4 if i == 0 else 0))   # stream 0: clean signal, 10 Hz off after sync entry
            iq = eng.tx(torch.tensor(feats[None], device=torch_dev))
            rx = eng.channel(iq, sigma, fo, n_pre=n_pre, n_post=1152, with_eoo=True, G=torch.tensor(G[None], device=torch_dev),
                             noise=torch.tensor(noise[None], device=torch_dev))
            f, st, _ = eng.rx(rx)
            out.append((rx.cpu().numpy()[0], f.cpu().numpy()[0, :st[0].n_valid], eng.rx_trace(0), (st[0].consumed, st[0].n_calls, st[0].n_valid, st[0].has_eoo, st[0].nin, st[0].state)))
            eng.close()
        res[mode] = out
    n_fail = 0
    for i, (a, bb) in enumerate(zip(res["1"], res[None])):
        for other in (a, res["rows3"][i]):
            assert other[3] == bb[3]
            for k in INT_KEYS:
                assert np.array_equal(other[2][k], bb[2][k]), (i, k)
            assert np.array_equal(other[1], bb[1])
        d = oracle.run_rx_stream(oracle_model, a[0], foff_err=10.0 if i == 0 else 0.0)
        for k in INT_KEYS:
            assert np.array_equal(bb[2][k], d[k]), (i, k)
        sa, sb = d["state_before"], d["state_after"]
        n_fail += int(np.sum((sa == 2) & (sb == 0)))
    assert n_fail >= 1            # the case really exercises sync losses


def _make_stream(seed, n_mf, EbNodB, fo, chan):
    from radae_amd.channel_tools import multipath_g, synth_features
    rng = np.random.default_rng(seed)
    feats = synth_features(seed, n_mf * 12)
    n_sig = n_mf * 960
    G = multipath_g(chan, 8000, n_sig, seed + 1) if chan != "awgn" else None
    n_pre = int(rng.integers(2000, 6000))
    n_tot = n_pre + n_sig + 1152 + 1152
    noise = ((rng.standard_normal(n_tot) + 1j * rng.standard_normal(n_tot)) / np.sqrt(2)).astype(np.complex64)
    return feats, G, n_pre, noise


def test_full_chain_vs_oracle_fresh_inputs(Engine, torch_dev, oracle, oracle_model, monkeypatch):
    """4 streams with different channels/offsets, inputs never seen by the golden generator."""
    import torch
    from radae_amd.engine import sigma_from_EbNodB
    n_mf = 16
    cases = [(21, 10.0, 11.0, "awgn"), (22, 6.0, -11.0, "mpp"), (23, 8.0, -28.0, "mpd"), (24, 10.0, 31.0, "mpg")]
    for seed, eb, fo, chan in cases:           # one engine per case: channel parameters are per call, not per stream
        feats, G, n_pre, noise = _make_stream(seed, n_mf, eb, fo, chan)
        sigma = sigma_from_EbNodB(eb)
        eng = Engine(1, max_tx_mf=n_mf, rx_trace_calls=48)
        iq = eng.tx(torch.tensor(feats[None], device=torch_dev))
        Gd = torch.tensor(G[None], device=torch_dev) if G is not None else None
        rx = eng.channel(iq, sigma, fo, n_pre=n_pre, n_post=1152, with_eoo=True, G=Gd, noise=torch.tensor(noise[None], device=torch_dev))
        fo_dev, st, _ = eng.rx(rx)
        t = eng.rx_trace(0)
        # oracle
        tx = oracle.Tx(oracle_model)
        sig = np.concatenate([tx.frame(feats[12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)])
        assert np.abs(iq.cpu().numpy()[0] - sig).max() < 5e-5
        r, fin = oracle.channel(sig, G, noise[n_pre:n_pre + len(sig)], sigma, fo)
        e = oracle.channel_eoo(tx.eoo(), noise[n_pre + len(sig):n_pre + len(sig) + 1152], sigma, fo, 0.0, fin)
        full = np.concatenate([sigma * noise[:n_pre], r, e, sigma * noise[-1152:]]).astype(np.complex64)
        assert np.abs(rx.cpu().numpy()[0] - full).max() < 5e-5
        d = oracle.run_rx_stream(oracle_model, full)
        for k in INT_KEYS:
            assert np.array_equal(t[k], d[k]), (chan, k)
        nv = st[0].n_valid
        assert nv == len(d["features_out"])
        if nv:
            assert rms(fo_dev.cpu().numpy()[0, :nv], d["features_out"]) < 1e-4
        eng.close()


def test_unbounded_operands_do_not_overflow(Engine, torch_dev, oracle, oracle_model):
    """The matrix products run on binary16 operand planes; the operands that are NOT tanh-bounded must not overflow them:
    raw features into the encoder's dense1 (+-300 and beyond), z_hat rows into the decoder's dense1 (+-1e3 / +-1e5: a deep
    fade or a false sync divides symbols by a tiny pilot magnitude), and the received samples themselves in check_pilots
    (int16-scaled input, bursts far above the pilots).  The reference computes finite values that tanh squashes
    (radae_base.py:260-268, :400-404); so must every path here -- small batches (f32 kernels), large ones (split-binary16
    kernels) and the decoder stage inside the receiver kernel."""
    import torch
    from radae_amd.channel_tools import synth_features
    from radae_amd.engine import sigma_from_EbNodB
    # (a) encoder, large batch (> 16 k GEMM rows) and single stream
    B, n_mf = 72, 84
    feats = np.stack([synth_features(900 + b, 12 * n_mf) for b in range(B)])
    feats[:, :, :20] *= np.float32(120.0)                     # |f0| up to ~1500, most features past +-256
    assert np.abs(feats).max() > 300
    eng = Engine(B, max_tx_mf=n_mf)
    z = eng.tx(torch.tensor(feats, device=torch_dev), want_z=True)[1].cpu().numpy()
    eng.close()
    assert np.isfinite(z).all()
    small = Engine(1, max_tx_mf=n_mf)
    for b in (0, 40):
        tx = oracle.Tx(oracle_model)
        zr = np.concatenate([tx.frame(feats[b, 12 * k:12 * k + 12].ravel())[1] for k in range(n_mf)]).reshape(-1, 80)
        assert np.abs(z[b] - zr).max() / np.abs(zr).max() < 2e-5
        zs = small.tx(torch.tensor(feats[b][None], device=torch_dev), want_z=True)[1].cpu().numpy()[0]
        assert np.abs(zs - zr).max() / np.abs(zr).max() < 2e-5
        small.tx_reset()
    small.close()
    # (b) stand-alone decoder, rows of +-1e3 and +-1e5
    rng = np.random.default_rng(77)
    zin = rng.standard_normal((B, 3 * n_mf, 80)).astype(np.float32)
    zin[:, 5::7] *= np.float32(1e3); zin[:, 9::11] *= np.float32(1e5)
    eng = Engine(B, max_tx_mf=n_mf)
    fh = eng.decode(torch.tensor(zin, device=torch_dev), 84).cpu().numpy()
    eng.close()
    assert np.isfinite(fh).all()
    small = Engine(1, max_tx_mf=n_mf)
    for b in (3, 60):
        dec = oracle.Decoder(oracle_model)
        ref = np.stack([dec.step(r) for r in zin[b]])
        assert rms(fh[b], ref) < 1e-4
        assert rms(small.decode(torch.tensor(zin[b][None], device=torch_dev), 84).cpu().numpy()[0], ref) < 1e-4
    small.close()
    # (c) the receiver: int16-scaled samples; a burst far above the pilots (check_pilots rows and the end-of-over correlation see
    # samples of ~2000); pilots faded out by 80 dB at 60 dB Eb/No, so that z_hat = symbol / pilot magnitude leaves +-256
    n_mf = 30
    f1, G, n_pre, noise = _make_stream(31, n_mf, 8.0, 7.0, "awgn")
    tx = oracle.Tx(oracle_model)
    sig = np.concatenate([tx.frame(f1[12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)])
    boosted = sig.copy(); faded = sig.copy()
    for mf in range(12, 18):                                  # data symbols (the four after the pilot symbol) x 2000 in six frames
        boosted[mf * 960 + 192:(mf + 1) * 960] *= np.float32(2000.0)
    for mf in range(12, 20):                                  # pilot symbols of eight frames x 1e-4
        faded[mf * 960:mf * 960 + 192] *= np.float32(1e-4)
    streams = []
    for s_in, gain, eb in ((sig, 3.0e4, 8.0), (boosted, 1.0, 8.0), (faded, 1.0, 60.0)):
        sigma = sigma_from_EbNodB(eb)
        r, fin = oracle.channel(s_in, None, noise[n_pre:n_pre + len(sig)], sigma, 7.0)
        e = oracle.channel_eoo(tx.eoo(), noise[n_pre + len(sig):n_pre + len(sig) + 1152], sigma, 7.0, 0.0, fin)
        streams.append((np.concatenate([sigma * noise[:n_pre], r, e, sigma * noise[-1152:]]) * np.float32(gain)).astype(np.complex64))
    eng = Engine(3, max_tx_mf=1, rx_trace_calls=64)
    fo_dev, st, _ = eng.rx(torch.tensor(np.stack(streams), device=torch_dev))
    for b in range(3):
        d = oracle.run_rx_stream(oracle_model, streams[b])
        t = eng.rx_trace(b)
        for k in INT_KEYS:
            assert np.array_equal(t[k], d[k]), (b, k)
        nv = st[b].n_valid
        assert nv == len(d["features_out"]) and nv > 5
        out = fo_dev.cpu().numpy()[b, :nv]
        assert np.isfinite(out).all() and rms(out, d["features_out"]) < 1e-4
    assert np.abs(eng.rx_trace(2)["z_all"]).max() > 256.0       # the faded pilots really produced out-of-range latents
    eng.close()


def test_rx_output_capacity_is_respected(Engine, torch_dev, golden, monkeypatch):
    """features_out rows are a capacity: a stream that has filled them pauses (consumed < available) instead of writing on;
    continuing with a fresh buffer gives the same frames as one big call."""
    import torch
    g = golden("rxtrace_awgn")
    x = torch.tensor(g["rx_in"][None], device=torch_dev)
    eng = Engine(1, max_tx_mf=1)
    full, st, _ = eng.rx(x)
    nv = st[0].n_valid
    assert nv == len(g["features_out"]) and nv > 6
    eng.rx_reset()
    guard = torch.full((1, 5 + 1, 432), 7.0, dtype=torch.float32, device=torch_dev)
    part, st1, _ = eng.rx(x, features_out=guard[:, :5])
    assert st1[0].n_valid == 5 and st1[0].consumed < x.shape[1] and bool((guard[:, 5] == 7.0).all())      # nothing past row 5
    rest, st2, _ = eng.rx(x[:, st1[0].consumed:].contiguous())
    assert st2[0].n_valid == nv - 5
    assert torch.equal(torch.cat([part[0, :5], rest[0, :nv - 5]]), full[0, :nv])
    eng.close()


def _random_utterance(oracle, oracle_model, seed, chan, eb, fo, n_mf=24):
    """one utterance of tools/parity_sweep.py: the oracle's transmitter and channel make the received samples"""
    from radae_amd.channel_tools import multipath_g, synth_features
    from radae_amd.engine import sigma_from_EbNodB
    r2 = np.random.default_rng(seed)
    feats = synth_features(seed, n_mf * 12); n_sig = n_mf * 960
    G = multipath_g(chan, 8000, n_sig, seed + 1) if chan != "awgn" else None
    n_pre = int(r2.integers(1000, 9000)); n_tot = n_pre + n_sig + 2304
    noise = ((r2.standard_normal(n_tot) + 1j * r2.standard_normal(n_tot)) / np.sqrt(2)).astype(np.complex64)
    sigma = sigma_from_EbNodB(eb)
    tx = oracle.Tx(oracle_model)
    sig = np.concatenate([tx.frame(feats[12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)])
    r, fin = oracle.channel(sig, G, noise[n_pre:n_pre + n_sig], sigma, fo)
    e = oracle.channel_eoo(tx.eoo(), noise[n_pre + n_sig:n_pre + n_sig + 1152], sigma, fo, 0.0, fin)
    return np.concatenate([sigma * noise[:n_pre], r, e, sigma * noise[-1152:]]).astype(np.complex64)


def _receiver_vs_oracle(Engine, torch_dev, oracle, oracle_model, full):
    """both receivers on the same samples -> (discrete outputs equal, fmax of every call the same double, features within 1e-4 RMS)"""
    import torch
    d = oracle.run_rx_stream(oracle_model, full)
    eng = Engine(1, max_tx_mf=1, rx_trace_calls=64)
    fo_dev, st, _ = eng.rx(torch.tensor(full[None], device=torch_dev))
    t = eng.rx_trace(0); nv = st[0].n_valid
    eng.close()
    disc = all(np.array_equal(t[k], d[k]) for k in INT_KEYS) and nv == len(d["features_out"])
    fmax_eq = disc and np.array_equal(t["fmax"], d["fmax"])
    feat_ok = disc and (nv == 0 or rms(fo_dev.cpu().numpy()[0, :nv], d["features_out"]) < 1e-4)
    return disc, fmax_eq, feat_ok


def test_randomised_receiver_sweep_vs_oracle(Engine, torch_dev, oracle, oracle_model, monkeypatch):
    """tools/parity_sweep.py as a test: random channel (AWGN / MPP / MPD / MPG), Eb/No -1..12 dB, offset +-40 Hz, noise prefix; the oracle makes
    the received samples, both receivers consume exactly those.  Every per-call discrete output must be equal, the frequency estimate of every call
    the SAME DOUBLE, and the features within 1e-4 RMS.  (Rounds 2-3 tolerated "refine() near-ties" here, three per cent of random utterances whose fmax
    moved by a grid step: they were the device's FMA-contracted fmax update, see test_refine_grid_length_follows_the_reference_doubles.)"""
    rng = np.random.default_rng(2027)
    bad, N = [], 32
    for case in range(N):
        seed = int(rng.integers(1, 1 << 30)); eb = float(rng.uniform(-1.0, 12.0)); fo = float(rng.uniform(-40.0, 40.0))
        chan = ["awgn", "mpp", "mpd", "mpg"][int(rng.integers(0, 4))]
        disc, fmax_eq, feat_ok = _receiver_vs_oracle(Engine, torch_dev, oracle, oracle_model, _random_utterance(oracle, oracle_model, seed, chan, eb, fo))
        if not (disc and fmax_eq and feat_ok):
            bad.append((case, seed, chan, eb, fo, disc, fmax_eq, feat_ok))
    assert not bad, bad


@pytest.mark.parametrize("seed,chan,eb,fo", [(927382851, "awgn", 1.8613418150778065, 7.307787791453919), (319558060, "mpp", 0.5046517495824037, 7.847441899881964)])
def test_refine_grid_length_follows_the_reference_doubles(Engine, torch_dev, oracle, oracle_model, seed, chan, eb, fo):
    """Two utterances of tools/parity_sweep.py (sweep seeds 99 / 12345) on which the receivers parted in round 4: in the synchronised state the reference
    searches np.arange(fmax - 1, fmax + 1, 0.1), which has 20 or 21 points depending on the LAST BIT of the double fmax = 0.9 fmax + 0.1 fhat
    (radae_rxe.py:202-206).  The device evaluated that update (and fhat = start + i delta) as FMAs -- one rounding instead of two --, now and then had a
    20-point grid where the reference had 21, and on these two utterances the 21st point won (timing estimate off by 1 resp. 6 samples from that call on,
    loss of sync later).  The same contraction was behind every "refine() near-tie" of rounds 2-3.  Now: every discrete output and every fmax bit-equal."""
    disc, fmax_eq, feat_ok = _receiver_vs_oracle(Engine, torch_dev, oracle, oracle_model, _random_utterance(oracle, oracle_model, seed, chan, eb, fo))
    assert disc and fmax_eq and feat_ok, (disc, fmax_eq, feat_ok)


def test_streams_are_independent_and_ragged(Engine, torch_dev, golden, monkeypatch):
    """Identical streams give bit-identical outputs whatever their slot; ragged / empty inputs are handled."""
    import torch
    g = golden("rxtrace_awgn")
    x = g["rx_in"]
    B = 5
    buf = np.zeros((B, len(x)), np.complex64)
    avail = np.array([len(x), len(x), 5000, 0, 959], np.int32)
    for b in range(B):
        buf[b, :avail[b]] = x[:avail[b]]
    eng = Engine(B, max_tx_mf=1, rx_trace_calls=64)
    feats, st, _ = eng.rx(torch.tensor(buf, device=torch_dev), n_avail=avail)
    assert st[0].n_calls == len(g["ret"]) and st[1].n_calls == st[0].n_calls
    assert torch.equal(feats[0], feats[1])
    assert st[2].n_calls == 5 and st[2].consumed == 4800        # whole calls only
    assert st[3].n_calls == 0 and st[3].consumed == 0 and st[3].nin == 960
    assert st[4].n_calls == 0
    t0, t2 = eng.rx_trace(0), eng.rx_trace(2)
    for k in INT_KEYS:
        assert np.array_equal(t0[k][:5], t2[k]), k
    eng.close()


@pytest.mark.parametrize("name,B", [("mpp", 96), ("foff", 96), ("mpp", 300)])
def test_many_identical_streams_agree(Engine, torch_dev, golden, name, B):
    """Race / slot-dependence guard: many copies of one golden stream (96: more workgroups than fit one XCD; 300: more than
    the chip runs at once) must come out bit-identical in every slot, twice in a row, and equal to the golden trace."""
    import torch
    g = golden("rxtrace_" + name)
    buf = torch.tensor(np.stack([g["rx_in"]] * B), device=torch_dev)
    eng = Engine(B, max_tx_mf=1, rx_trace_calls=64, flags=4 if name == "foff" else 0)
    for rep in range(2):
        eng.rx_reset()
        f, st, _ = eng.rx(buf)
        f = f.cpu().numpy()
        for b in range(1, B):
            assert st[b].n_calls == st[0].n_calls and st[b].n_valid == st[0].n_valid, (rep, b)
            assert np.array_equal(f[b], f[0]), (rep, b)
        for b in (0, B // 2, B - 1):
            t = eng.rx_trace(b)
            for k in INT_KEYS:
                assert np.array_equal(t[k], g[k]), (rep, b, k)
    eng.close()


def test_replicas_agree_over_many_fresh_launches(Engine, torch_dev):
    """Barrier-phase race guard.  256 streams = 32 replicas of 8 utterances, a fresh engine per repetition, no trace (the timing of
    the untraced kernel is what exposed it): every replica must come out bit-identical.  Round 2 found a latent race here -- every
    thread evaluated the sync-entry condition from LDS scalars while thread 0 was already incrementing valid_count in its state
    update; a wavefront that read the new count entered refine() alone one call early, its barriers paired with the wrong ones,
    and it read the next call's samples (about 1 launch in 20 had such a stream; tools/first_launch_check.py is the long form)."""
    import torch
    from radae_amd.channel_tools import synth_features
    from radae_amd.engine import sigma_from_EbNodB
    B, T = 256, 1008
    n_mf = T // 12
    base = [synth_features(3000 + u, T) for u in range(8)]
    feats = torch.tensor(np.stack([base[b % 8] for b in range(B)]), device=torch_dev)
    rng = np.random.default_rng(9)
    n_tot = 4000 + n_mf * 960 + 1152 + 1152
    nz = ((rng.standard_normal((8, n_tot)) + 1j * rng.standard_normal((8, n_tot))) / np.sqrt(2)).astype(np.complex64)
    noise = torch.tensor(np.concatenate([nz] * 32), device=torch_dev)
    for rep in range(40):
        eng = Engine(B, max_tx_mf=n_mf)
        rx = eng.channel(eng.tx(feats), sigma_from_EbNodB(10.0), 11.0, n_pre=4000, n_post=1152, with_eoo=True, noise=noise)
        fo, st, _ = eng.rx(rx)
        nv = np.array([s.n_valid for s in st])
        assert nv.min() == nv.max() == 81, (rep, nv.min(), nv.max())
        assert torch.equal(fo.view(32, 8, -1)[1:], fo.view(32, 8, -1)[:1].expand(31, -1, -1)), rep
        eng.close()


def test_tx_frame_three_rows_equals_step_kernel(tmp_path):
    """rade_tx as one launch that takes the frame's three encoder steps through each layer together (k_tx_frame3) against the form that runs the step
    three times (k_tx_frame, $RADE_TX_FRAME_BY_STEP=1): same chunk -> wavefront assignment, same order of partial sums per row -- the transmit
    samples of 20 consecutive frames (state carried from frame to frame) are equal bit for bit.  Two processes: the switch is read once per process."""
    import subprocess, sys
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from radae_amd import api\n"
        "from radae_amd.channel_tools import synth_features\n"
        "f = synth_features(4242, 240); tx = api.radae_tx(); out = np.zeros((20, 960), np.complex64)\n"
        "for k in range(20): tx.do_radae_tx(f[12 * k:12 * k + 12].ravel(), out[k])\n"
        "np.save(sys.argv[1], out)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, extra in (("three", {}), ("step", {"RADE_TX_FRAME_BY_STEP": "1"})):
        f = str(tmp_path / (name + ".npy"))
        env = dict(os.environ); env.pop("RADE_TX_FRAME_BY_STEP", None); env.update(extra)
        subprocess.run([sys.executable, "-c", script, f], check=True, env=env, timeout=300)
        outs.append(np.load(f))
    assert np.abs(outs[0]).max() > 0.1
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_single_stream_c_abi(golden):
    """rade_api.h entry points (what radae_tx.c / radae_rx.c / freedv-gui call), via radae_amd.api."""
    from radae_amd import api
    e = golden("enc_tx"); c = golden("consts"); g = golden("rxtrace_awgn")
    h = api.Rade()
    L = h.L
    assert L.rade_version() == 1
    assert (L.rade_n_tx_out(h.r), L.rade_n_tx_eoo_out(h.r), L.rade_nin_max(h.r), L.rade_n_features_in_out(h.r), L.rade_n_eoo_bits(h.r)) == (960, 1152, 1120, 432, 180)
    assert L.rade_freq_offset(h.r) == 0.0 and L.rade_sync(h.r) == 0
    tx = api.radae_tx(handle=h)
    out = np.zeros(960, np.complex64)
    for k in range(10):
        tx.do_radae_tx(e["features"][0, 12 * k:12 * k + 12].ravel(), out)
        assert np.abs(out - e["tx"][0, k]).max() < 2e-5
    eo = np.zeros(1152, np.complex64)
    tx.do_eoo(eo); assert np.abs(eo - c["eoo_default"]).max() < 1e-6
    tx.set_eoo_bits(c["eoo_bits_in"]); tx.do_eoo(eo); assert np.abs(eo - c["eoo_with_bits"]).max() < 1e-6
    rx = api.radae_rx(handle=h)
    x = g["rx_in"]; pos = 0; rets, nins, feats, snrs = [], [], [], []
    fo = np.zeros(432, np.float32)
    while pos + rx.get_nin() <= len(x):
        nin = rx.get_nin()
        buf = np.zeros(rx.get_nin_max(), np.complex64); buf[:nin] = x[pos:pos + nin]; pos += nin
        r = rx.do_radae_rx(buf, fo)
        rets.append(r); nins.append(rx.get_nin()); snrs.append(rx.get_snrdB_3k_est())
        if r & 1:
            feats.append(fo.copy())
        if r & 2:
            assert np.abs(fo[:180] - g["eoo_out"][-1]).max() < 1e-4
    assert np.array_equal(rets, g["ret"]) and np.array_equal(nins, g["nin_after"]) and np.array_equal(snrs, g["snr_int"])
    assert rms(np.array(feats), g["features_out"]) < 1e-5
    h.close()


def test_full_size_batch_properties(Engine, torch_dev, oracle, oracle_model, monkeypatch):
    """BASELINE workload size (256 x 1008 frames): properties that do not need the oracle at full size,
    plus the oracle on two of the streams (loss delta < 1e-4)."""
    import torch
    from radae_amd.channel_tools import synth_features
    from radae_amd.engine import sigma_from_EbNodB
    from radae_amd.loss import find_loss
    B, T = 256, 1008
    n_mf = T // 12
    base = [synth_features(3000 + u, T) for u in range(8)]
    feats = np.stack([base[b % 8] for b in range(B)])                 # 8 distinct utterances, replicated 32x
    eng = Engine(B, max_tx_mf=n_mf, rx_trace_calls=0)
    iq = eng.tx(torch.tensor(feats, device=torch_dev))
    assert torch.equal(iq[:8], iq[248:256])                           # replicas bit-identical
    mag = iq.abs()
    assert float(mag.max()) <= 1.0 + 1e-6                             # tanh PA limiter
    papr = 20 * np.log10(float(mag.max()) / float(torch.sqrt((mag ** 2).mean())))
    assert papr < 1.0                                                 # README.md:434 "PAPR < 1 dB"
    rng = np.random.default_rng(9)
    n_tot = 4000 + n_mf * 960 + 1152 + 1152
    nz = ((rng.standard_normal((8, n_tot)) + 1j * rng.standard_normal((8, n_tot))) / np.sqrt(2)).astype(np.complex64)
    noise = torch.tensor(np.concatenate([nz] * 32), device=torch_dev)
    sigma = sigma_from_EbNodB(10.0)
    rx = eng.channel(iq, sigma, 11.0, n_pre=4000, n_post=1152, with_eoo=True, noise=noise)
    fo, st, eoo = eng.rx(rx)
    nv = np.array([s.n_valid for s in st])
    assert nv.min() >= n_mf - 4 and all(s.has_eoo for s in st)        # every stream syncs and sees the end of over
    assert torch.equal(fo[:8], fo[248:256])
    for b in (0, 5):
        tx = oracle.Tx(oracle_model)
        sig = np.concatenate([tx.frame(feats[b, 12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)])
        r, fin = oracle.channel(sig, None, nz[b, 4000:4000 + len(sig)], sigma, 11.0)
        e = oracle.channel_eoo(tx.eoo(), nz[b, 4000 + len(sig):4000 + len(sig) + 1152], sigma, 11.0, 0.0, fin)
        full = np.concatenate([sigma * nz[b, :4000], r, e, sigma * nz[b, -1152:]]).astype(np.complex64)
        d = oracle.run_rx_stream(oracle_model, full)
        assert len(d["features_out"]) == nv[b]
        got = fo[b, :nv[b]].cpu().numpy()
        assert rms(got, d["features_out"]) < 1e-4
        l_o, s_o = find_loss(feats[b], d["features_out"].reshape(-1, 36))
        l_g, s_g = find_loss(feats[b], got.reshape(-1, 36))
        assert s_o == s_g and abs(l_o - l_g) < 1e-4                   # loss.py delta vs the oracle < 1e-4
    eng.close()


def test_config3_full_size_mpp_vs_oracle(Engine, torch_dev, oracle, oracle_model):
    """BASELINE.json configs[2] at full size inside `pytest -m gpu`, through the call bench.py times (rade_batch_tx_channel, one pass): 256 streams x 1008
    frames, MPP Doppler-spread two-path channel (G resident on the device), AWGN at Eb/No = 3 dB, -11 Hz, 1 s of noise in front, EOO frame + 1152 samples behind.
    The noise is an explicit tensor here (bench.py: device Philox), so that the oracle can be run on two of the streams: per-call discrete outputs equal,
    features < 1e-4 RMS, loss.py delta < 1e-4."""
    import torch
    from radae_amd.channel_tools import multipath_g, synth_features
    from radae_amd.engine import sigma_from_EbNodB
    from radae_amd.loss import find_loss
    B, T = 256, 1008
    n_mf = T // 12; n_sig = n_mf * 960; n_pre, n_post = 8000, 1152
    nd = 8                                                            # distinct utterances / channels / noises, replicated 32x
    feats = np.stack([synth_features(1000 + (b % nd), T) for b in range(B)])
    Gs = [multipath_g("mpp", 8000, n_sig, 5000 + u) for u in range(nd)]
    G = torch.tensor(np.stack([Gs[b % nd] for b in range(B)]), device=torch_dev)
    rng = np.random.default_rng(2026)
    n_tot = n_pre + n_sig + 1152 + n_post
    nz = ((rng.standard_normal((nd, n_tot)) + 1j * rng.standard_normal((nd, n_tot))) / np.sqrt(2)).astype(np.complex64)
    noise = torch.tensor(np.concatenate([nz] * (B // nd)), device=torch_dev)
    sigma = sigma_from_EbNodB(3.0)
    eng = Engine(B, max_tx_mf=n_mf, rx_trace_calls=128)
    rx, iq = eng.tx_channel(torch.tensor(feats, device=torch_dev), sigma, -11.0, n_pre=n_pre, n_post=n_post, with_eoo=True, G=G, noise=noise, want_iq=True)
    fo, st, _ = eng.rx(rx)
    nv = np.array([s.n_valid for s in st])
    assert torch.equal(rx[:nd], rx[B - nd:]) and torch.equal(fo[:nd], fo[B - nd:])          # replicas bit-identical
    assert nv.mean() > 0.8 * n_mf                                     # MPP at 3 dB: most frames are decoded (fades cost a few)
    keys = ["state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]
    for b in (1, 6):
        tx = oracle.Tx(oracle_model)
        sig = np.concatenate([tx.frame(feats[b, 12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)])
        assert np.abs(iq[b].cpu().numpy() - sig).max() < 5e-5
        r, fin = oracle.channel(sig, Gs[b], nz[b, n_pre:n_pre + n_sig], sigma, -11.0)
        e = oracle.channel_eoo(tx.eoo(), nz[b, n_pre + n_sig:n_pre + n_sig + 1152], sigma, -11.0, 0.0, fin)
        full = np.concatenate([sigma * nz[b, :n_pre], r, e, sigma * nz[b, -n_post:]]).astype(np.complex64)
        got_rx = rx[b].cpu().numpy()
        assert rms(got_rx, full) < 2e-5
        d = oracle.run_rx_stream(oracle_model, got_rx)               # the receiver on the SAME samples the device receiver read
        t = eng.rx_trace(b)
        for k in keys:
            assert np.array_equal(t[k][:len(d[k])], d[k]), (b, k)
        assert len(d["features_out"]) == nv[b] and nv[b] > 60
        got = fo[b, :nv[b]].cpu().numpy()
        assert rms(got, d["features_out"]) < 1e-4
        l_o, s_o = find_loss(feats[b], d["features_out"].reshape(-1, 36))
        l_g, s_g = find_loss(feats[b], got.reshape(-1, 36))
        assert s_o == s_g and abs(l_o - l_g) < 1e-4                   # loss.py delta vs the oracle < 1e-4
    eng.close()


def test_device_multipath_generator(Engine, torch_dev):
    """the Watterson / Doppler-spread generator on the device.  With the host generator's own low-rate
    noise as input it reproduces radae_amd.channel_tools.multipath_g (FIR, interpolation, hf_gain); from its Philox noise
    the statistics are right: var G1 + var G2 = 1 per stream and the autocorrelation exp(-(pi sigma tau)^2) of a process whose filter has the Gaussian
    AMPLITUDE response exp(-f^2 / (2 sigma^2)) (doppler_spread.m:20-27).  The host generator's taps are fir2's recipe pinned on scipy.signal.firwin2
    (tests/test_host_cpu.py::test_doppler_filter_design_against_scipy_firwin2)."""
    import torch
    from radae_amd.channel_tools import multipath_g, doppler_plan, PRESETS
    B, n = 6, 40 * 960
    eng = Engine(B, max_tx_mf=1)
    for ch in ("mpp", "mpd"):
        taps, ratio, n_low = doppler_plan(PRESETS[ch][0], 8000, n)
        noise = np.zeros((B, 2, n_low + len(taps)), np.complex64); ref = []
        for b in range(B):
            rng = np.random.default_rng(40 + b)
            for p in range(2):                              # the draw order of doppler_spread(): real block, then imaginary block, per path
                noise[b, p] = rng.standard_normal(n_low + len(taps)) + 1j * rng.standard_normal(n_low + len(taps))
            ref.append(multipath_g(ch, 8000, n, 40 + b))
        G = eng.multipath_gen(ch, n, noise_low=torch.tensor(noise, device=torch_dev)).cpu().numpy()
        assert np.abs(G - np.stack(ref)).max() < 2e-5, ch     # complex64 input noise vs the host's float64 draw
    Bs, n = 6, 84 * 960
    G = eng.multipath_gen("mpp", n, seed=1234).cpu().numpy().astype(np.complex128)
    for b in range(Bs):
        assert abs(np.var(G[b, :, 0]) + np.var(G[b, :, 1]) - 1.0) < 1e-5
    sigma = PRESETS["mpp"][0] / 2.0
    for tau, tol in ((0.1, 0.05), (0.5, 0.2)):
        lag = int(tau * 8000)
        num = np.mean([np.real(np.vdot(G[b, :-lag, p], G[b, lag:, p])) / np.real(np.vdot(G[b, :, p], G[b, :, p])) for b in range(Bs) for p in range(2)])
        assert abs(num - np.exp(-(np.pi * sigma * tau) ** 2)) < tol, (tau, num)
    assert np.abs(G[0] - G[1]).max() > 0.1                 # streams are independent
    eng.close()


def test_lmr60_rate_rs_channel_matrix(Engine, torch_dev):
    """multipath_samples.m:17-21 + :33-40 (BBFM.md:37: `multipath_samples("lmr60", 8000, 2000, 1, 10, "h_lmr60.f32")`): the land-mobile preset (60 km/h at 450 MHz: 50 Hz spread,
    200 us) and the rate-Rs |H| the script derives from the rate-Fs Doppler samples, generated on the device.  10 s = 5334 low-rate points, more than the generator
    keeps in LDS (HBM scratch path).  With the host's noise as input it reproduces channel_tools.multipath_h; from Philox noise mean |H|^2 ~ 1 and the level-crossing
    rate the script itself checks (:48-61) is near sqrt(2 pi P / Pav) fd exp(-P / Pav)."""
    import torch
    from radae_amd.channel_tools import PRESETS, doppler_plan, multipath_h
    B, n_sym = 3, 20000
    n_g = (n_sym - 1) * 4 + 1
    taps, ratio, n_low = doppler_plan(PRESETS["lmr60"][0], 8000, n_g)
    assert ratio == 15 and n_low > 2048        # 2 * 450e6 * (60e3 / 3600 / 3e8) = 50.00000000000001 in doubles (Octave's too): lowFs = ceil(500.0000000000001) = 501 -> M = floor(8000 / 501) = 15
    eng = Engine(B, max_tx_mf=1)
    noise = np.zeros((B, 2, n_low + len(taps)), np.complex64); ref = []
    for b in range(B):
        rng = np.random.default_rng(60 + b)
        for p in range(2):
            noise[b, p] = rng.standard_normal(n_low + len(taps)) + 1j * rng.standard_normal(n_low + len(taps))
        ref.append(multipath_h("lmr60", 8000, 2000, 1, n_sym, 60 + b))
    H = eng.multipath_h_gen("lmr60", n_sym, noise_low=torch.tensor(noise, device=torch_dev)).cpu().numpy()
    assert H.shape == (B, n_sym, 1) and np.abs(H - np.stack(ref)).max() < 3e-5
    # several carriers, complex form (the rate-Rs RADE model's H: Rs = 50, Nc = 20 would need Fs / Rs = 160; here the BBFM rates with Nc = 3 to exercise the phase term)
    Hc = eng.multipath_h_gen("lmr60", 2000, nc=3, noise_low=None, seed=5, complex_=True).cpu().numpy()
    G = eng.multipath_gen("lmr60", 1999 * 4 + 1, seed=5).cpu().numpy()
    want = G[:, ::4, 0][:, :, None] + G[:, ::4, 1][:, :, None] * np.exp(-2j * np.pi * np.arange(3)[None, None, :] * 200e-6 * 2000)
    assert np.abs(Hc - want).max() < 1e-5
    Hp = eng.multipath_h_gen("lmr60", n_sym, seed=77).cpu().numpy()[:, :, 0].astype(np.float64)
    for b in range(B):
        pav = np.mean(Hp[b] ** 2)
        assert 0.85 < pav < 1.15
        lcr = np.sum((Hp[b, :-1] ** 2 < 1.0) & (Hp[b, 1:] ** 2 > 1.0)) / 10.0
        th = np.sqrt(2 * np.pi / pav) * 25.0 * np.exp(-1.0 / pav)
        assert 0.75 * th < lcr < 1.25 * th, (lcr, th)
    eng.close()


def test_channel_sine_interferer_and_gain(Engine, torch_dev, golden):
    """inference.py:285-289: --sine_amp/--sine_freq add a complex tone over the whole output, --rx_gain scales it."""
    import torch
    e = golden("enc_tx")
    eng = Engine(2, max_tx_mf=10)
    iq = eng.tx(torch.tensor(e["features"], device=torch_dev))
    base = eng.channel(iq, 0.0, 0.0, n_pre=100, n_post=50).cpu().numpy()
    out = eng.channel(iq, 0.0, 0.0, n_pre=100, n_post=50, sine_amp=0.3, sine_freq=1234.5, rx_gain=0.5).cpu().numpy()
    nidx = np.arange(base.shape[1])
    want = 0.5 * (base + 0.3 * np.exp(1j * nidx * 2 * np.pi * 1234.5 / 8000.0)[None])
    assert np.abs(out - want).max() < 2e-6
    eng.close()


def test_impulses_in_the_receive_buffer(Engine, torch_dev, golden, oracle, oracle_model, monkeypatch):
    """Dynamic range: check_pilots (both kernels) and the pilot search of k_rx_sync2 feed the matrix cores with rx_buf in two binary16 planes
    under ONE power-of-two scale taken from the running maximum of the buffer, so a click 50 .. 90 dB above the signal costs the signal that many
    bits of the 22.  The MPP golden input with two impulses of 300x / 30000x its RMS (in the noise prefix, and inside the synchronised part)
    must still give the oracle's discrete outputs call by call and its features."""
    import torch
    base = golden("rxtrace_mpp")["rx_in"].astype(np.complex64)
    r = float(np.sqrt(np.mean(np.abs(base) ** 2)))
    for amp, pos in [(300.0, 5000), (300.0, 20000), (30000.0, 5000), (30000.0, 20000)]:
        x = base.copy(); x[pos] += amp * r * (1 + 1j) / np.sqrt(2); x[pos + 700] -= amp * r
        d = oracle.run_rx_stream(oracle_model, x)
        eng = Engine(1, max_tx_mf=1, rx_trace_calls=64)
        fo, st, _ = eng.rx(torch.tensor(x[None], device=torch_dev))
        t = eng.rx_trace(0); nv = st[0].n_valid
        eng.close()
        for k in INT_KEYS:
            assert np.array_equal(t[k], d[k]), (amp, pos, k)
        assert nv == len(d["features_out"]) and nv > 0
        assert rms(fo.cpu().numpy()[0, :nv], d["features_out"]) < 1e-5, (amp, pos)


@pytest.mark.parametrize("kind", ["noise", "sine"])
def test_must_not_acquire(Engine, torch_dev, oracle, oracle_model, kind, monkeypatch):
    """The reference's acq_noise / acq_sine ctests (CMakeLists.txt:191-208): real-valued noise, or a 1 kHz sine in noise,
    converted with Q = 0 (int16tof32.py --zeropad), must never synchronise.  12 s per stream, 8 streams with different
    seeds; stream 0 is also checked call by call against the oracle."""
    import torch
    B, n = 8, 96000
    rng = np.random.default_rng(77 if kind == "noise" else 78)
    x = 0.1 * rng.standard_normal((B, n))
    if kind == "sine":
        x += 0.25 * np.cos(2 * np.pi * 1000.0 / 8000.0 * np.arange(n))[None]
    rx = x.astype(np.float32).astype(np.complex64)                     # Q == 0
    eng = Engine(B, max_tx_mf=1, rx_trace_calls=128)
    f, st, _ = eng.rx(torch.tensor(rx, device=torch_dev))
    for b in range(B):
        assert st[b].n_calls == n // 960 and st[b].n_valid == 0 and st[b].sync == 0, (kind, b)
        assert not np.any(eng.rx_trace(b)["state_after"] == 2), (kind, b)
    t = eng.rx_trace(0)
    d = oracle.run_rx_stream(oracle_model, rx[0])
    for k in INT_KEYS:
        assert np.array_equal(t[k], d[k]), (kind, k)
    assert np.abs(t["Dtmax12"] - d["Dtmax12"]).max() < 3e-5 * max(1.0, np.abs(d["Dtmax12"]).max())
    eng.close()


def test_acquisition_statistics_mpp(Engine, torch_dev, monkeypatch):
    """rx.py --acq_test in batch form (rx.py:163-195, ctest acq_mpp): 64 utterances at 0 dB Eb/No on the MPP channel with
    a +10 Hz offset.  Every stream must find sync, in less than 1.5 s of signal on average, with the entry timing inside
    the 2.5 ms window and the coarse frequency within 5 Hz of the truth for at least 80 % of the streams."""
    import torch
    from radae_amd.engine import sigma_from_EbNodB
    from radae_amd.channel_tools import multipath_g, synth_features
    B, n_mf, n_pre, fo = 64, 40, 4000, 10.0
    feats = np.stack([synth_features(300 + b, 12 * n_mf) for b in range(B)])
    G = np.stack([multipath_g("mpp", 8000, n_mf * 960, 900 + b) for b in range(B)])
    eng = Engine(B, max_tx_mf=n_mf, rx_trace_calls=64)
    iq = eng.tx(torch.tensor(feats, device=torch_dev))
    rx = eng.channel(iq, sigma_from_EbNodB(0.0), fo, n_pre=n_pre, n_post=1152, with_eoo=True, G=torch.tensor(G, device=torch_dev), seed=5)
    f, st, _ = eng.rx(rx)
    acq_calls, ok = [], 0
    for b in range(B):
        t = eng.rx_trace(b)
        entry = np.nonzero((t["state_before"] == 1) & (t["state_after"] == 2))[0]
        assert len(entry) > 0, b
        c = int(entry[0])
        acq_calls.append(c)
        # modem frame k starts at sample n_pre + 960 k; its pilot body follows the 32-sample cyclic prefix and leaves the
        # band-pass filter 50 + 2 samples later (dsp.py:55 vs :96); rx_buf holds the last 2112 samples after c + 1 calls
        t_true = (n_pre + 32 + 52 - (960 * (c + 1) - 2112)) % 960
        dt = (int(t["tmax"][c]) - t_true + 480) % 960 - 480
        ok += int(-20 < dt < 20 + 16 and abs(float(t["fmax"][c]) - fo) <= 5.0)      # 2.5 ms window (rx.py:176); the MPP echo is 16 samples late
    mean_t = (np.mean(acq_calls) + 1) * 0.12 - n_pre / 8000.0
    assert ok >= 0.8 * B, (ok, B)
    assert mean_t < 1.5, mean_t
    eng.close()


def _pipe(exe, args, data, cwd):
    import subprocess
    p = subprocess.run([exe] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=cwd, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout


@pytest.mark.parametrize("which", ["own", "reference"])
def test_stdin_stdout_hosts(golden, tmp_path, which):
    """features.f32 | radae_tx > iq.f32 and iq.f32 | radae_rx > features.f32 over the C ABI: our own hosts
    (hosts/) and, when built, the reference's radae_tx.c / radae_rx.c compiled unmodified (oracle/_ref/)."""
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if which == "own":
        txe, rxe, args = os.path.join(repo, "hosts", "rade_tx_filter"), os.path.join(repo, "hosts", "rade_rx_filter"), [""]
    else:
        txe, rxe, args = os.path.join(repo, "oracle", "_ref", "radae_tx"), os.path.join(repo, "oracle", "_ref", "radae_rx"), []
    if not (os.path.exists(txe) and os.path.exists(rxe)):
        pytest.skip(f"{txe} not built")
    e = golden("enc_tx"); c = golden("consts"); g = golden("rxtrace_awgn")
    c["eoo_bits_in"].astype(np.float32).tofile(tmp_path / "eoo_tx.f32")
    iq = np.frombuffer(_pipe(txe, args, e["features"][0].astype(np.float32).tobytes(), tmp_path), np.complex64)
    assert len(iq) == 10 * 960 + 2 * 1152
    assert np.abs(iq[:9600].reshape(10, 960) - e["tx"][0]).max() < 2e-5
    assert np.abs(iq[9600:9600 + 1152] - c["eoo_with_bits"]).max() < 1e-6 and not iq[9600 + 1152:].any()
    feats = np.frombuffer(_pipe(rxe, args, g["rx_in"].astype(np.complex64).tobytes(), tmp_path), np.float32).reshape(-1, 432)
    assert feats.shape == g["features_out"].shape and rms(feats, g["features_out"]) < 1e-5
    eoo = np.fromfile(tmp_path / "eoo_rx.f32", np.float32)
    assert eoo.shape == (180,) and np.abs(eoo - g["eoo_out"][-1]).max() < 1e-4


def test_core_level_boundary_config2(golden, tmp_path):
    """include/rade_core.h (rade_core.h:42-46 of the reference): rade_core_encoder / rade_core_decoder one 40 ms step per call
    against the goldens of the reference's stateful modules, bottleneck 1 = tanh of the same latents, a second state on the
    same model is independent, and the stdin/stdout filters with the test_rade_enc.c / test_rade_dec.c command lines and wire
    formats (84-float rows <-> 80-float rows inside 4 x 36-float frames)."""
    import os
    from radae_amd import core
    e, d = golden("enc_tx"), golden("dec_loss")
    f = e["features"][0]                                               # [120][36]
    rows = np.concatenate([f[:, :20], -np.ones((120, 1), np.float32)], 1).reshape(30, 84)
    enc = core.CoreEncoder()
    z = np.stack([enc.step(r) for r in rows])
    assert rms(z, e["z"][0]) < 1e-4 and np.abs(z - e["z"][0]).max() < 2e-6 * np.abs(e["z"]).max() + 1e-5     # latents are O(100)
    enc2 = core.CoreEncoder()                                          # second state, same blob: bottleneck 1, untouched by the first
    z1 = np.stack([enc2.step(r, bottleneck=1) for r in rows])
    assert np.abs(z1 - np.tanh(e["z"][0])).max() < 1e-4
    enc.reset()
    assert np.array_equal(np.stack([enc.step(r) for r in rows[:5]]), z[:5])          # rade_init_encoder == fresh state
    with pytest.raises(ValueError):
        core.CoreEncoder(input_dim=80)                                # model19_check3 carries the aux symbol: 84 only
    dec = core.CoreDecoder()
    fh = np.stack([dec.step(r) for r in d["z_hat"]]).reshape(120, 21)
    assert rms(fh, d["features"]) < 1e-5
    enc.close(); enc2.close(); dec.close()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = os.path.join(repo, "weights", "model19_check3.bin")
    zf = np.frombuffer(_pipe(os.path.join(repo, "hosts", "rade_enc_filter"), ["3", "1", blob], f.astype(np.float32).tobytes(), tmp_path), np.float32).reshape(-1, 80)
    assert np.array_equal(zf, z)                                       # the filter is the same call sequence
    ff = np.frombuffer(_pipe(os.path.join(repo, "hosts", "rade_dec_filter"), ["1"], d["z_hat"].astype(np.float32).tobytes(), tmp_path), np.float32).reshape(-1, 36)
    assert ff.shape == (120, 36) and np.array_equal(ff[:, :21], fh) and not ff[:, 21:].any()


@pytest.mark.parametrize("pipeline", [1, 3])
def test_multi_gpu_c_host_single_device(tmp_path, pipeline):
    """The C multi-GPU host (hosts/rade_multi_bench over rade_multi_*: RCCL blob broadcast, sharded engines, RCCL all-reduce of the
    statistics) on the one GPU this box has, with the RCCL path forced: the communicator, the broadcast and the all-reduce really run."""
    import json, os, subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "hosts", "rade_multi_bench")
    env = dict(os.environ, RADE_MULTI_FORCE_RCCL="1")
    p = subprocess.run([exe, "--gpus", "1", "--streams-per-gpu", "12", "--frames", "240", "--steps", "4", "--warmup", "1", "--pipeline", str(pipeline),
                        os.path.join(repo, "weights", "model19_check3.bin")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["collectives"].startswith("rccl") and line["config"]["batches_in_flight_per_gpu"] == pipeline
    j = line["job_last_step"]
    assert j["offered_frames"] == 12 * 240 and 0 < j["decoded_frames"] <= j["offered_frames"] and j["rx_calls"] >= 12 * 20
    assert j["samples_consumed"] > 12 * 20 * 800 and line["value"] > 0


def test_model05_rate_rs_config1(Engine, torch_dev, golden):
    """BASELINE config 1 on the GPU: model05 blob, bottleneck 1, rate-Rs symbol channel, stand-alone decoder."""
    import os
    import torch
    from radae_amd.engine import DEFAULT_BLOB
    from radae_amd.loss import distortion_loss
    g = golden("model05")
    blob = os.path.join(os.path.dirname(DEFAULT_BLOB), "model05.bin")
    T = g["features"].shape[0] // 4
    eng = Engine(1, max_tx_mf=T // 3, blob=blob, flags=0x100)          # RADE_BATCH_BOTTLENECK1
    z = eng.encode(torch.tensor(g["features"].reshape(1, T, 80), device=torch_dev))
    assert rms(z.cpu().numpy()[0], g["z"]) < 2e-5
    for tag in ("awgn", "mp"):
        zh = eng.channel_symbol(torch.tensor(g["z"][None], device=torch_dev), "rs", float(g[tag + "_sigma"]),
                                H=torch.tensor(g[tag + "_H"].reshape(1, -1), device=torch_dev), noise=torch.tensor(g[tag + "_noise"][None], device=torch_dev))
        assert np.abs(zh.cpu().numpy()[0] - g[tag + "_z_hat"]).max() < 1e-6
        fh = eng.decode(torch.tensor(g[tag + "_z_hat"][None], device=torch_dev), 80).cpu().numpy()[0].reshape(-1, 20)
        assert rms(fh, g[tag + "_features_hat"]) < 2e-5
        assert abs(distortion_loss(g["features"], fh) - float(g[tag + "_loss"])) < 1e-4
    # chunked == whole (stateful == stateless, ctests stateful_encoder / stateful_decoder)
    eng.tx_reset()
    f = torch.tensor(g["features"].reshape(1, T, 80), device=torch_dev)
    zc = torch.cat([eng.encode(f[:, a:b].contiguous()) for a, b in ((0, 1), (1, 4), (4, 100), (100, T))], 1)
    assert rms(zc.cpu().numpy()[0], z.cpu().numpy()[0]) < 1e-5
    zt = torch.tensor(g["awgn_z_hat"][None], device=torch_dev)
    whole = eng.decode(zt, 80)
    parts = torch.cat([eng.decode(zt[:, a:b].contiguous(), 80, reset=(a == 0)) for a, b in ((0, 3), (3, 4), (4, T))], 1)
    assert rms(parts.cpu().numpy(), whole.cpu().numpy()) < 1e-5
    eng.close()


def test_bbfm_config5(Engine, torch_dev, golden):
    """BASELINE config 5 on the GPU: encoder (bottleneck 1) -> FM-demodulator SNR channel -> decoder."""
    import os
    import torch
    from radae_amd.engine import DEFAULT_BLOB
    g = golden("bbfm")
    blob = os.path.join(os.path.dirname(DEFAULT_BLOB), "bbfm_random_seed20240501.bin")
    T = g["features"].shape[0] // 4
    eng = Engine(1, max_tx_mf=T // 3, blob=blob, flags=0x100)
    z = eng.encode(torch.tensor(g["features"].reshape(1, T, 80), device=torch_dev))
    assert rms(z.cpu().numpy()[0], g["z"]) < 1e-5
    for tag in ("awgn", "ray"):
        zh = eng.channel_symbol(torch.tensor(g["z"][None], device=torch_dev), "bbfm", float(g[tag + "_CNRdB"]), float(g["Gfm"]),
                                H=torch.tensor(g[tag + "_H"][None], device=torch_dev), noise=torch.tensor(g[tag + "_noise"][None], device=torch_dev))
        assert np.abs(zh.cpu().numpy()[0] - g[tag + "_z_hat"]).max() < 5e-6
        fh = eng.decode(torch.tensor(g[tag + "_z_hat"][None], device=torch_dev), 80).cpu().numpy()[0].reshape(-1, 20)
        assert rms(fh, g[tag + "_features_hat"]) < 1e-5
    # on-chip noise: same statistics as the reference's sigma (bbfm.py:182-184), batch of 256 streams
    eng.close()
    B = 256
    eng = Engine(B, max_tx_mf=T // 3, blob=blob, flags=0x100)
    zz = torch.zeros((B, T, 80), device=torch_dev)
    zh = eng.channel_symbol(zz, "bbfm", 20.0, float(g["Gfm"]), seed=7)
    sig = 10 ** (-(20.0 + float(g["Gfm"])) / 20)
    assert float(zh.std()) == pytest.approx(sig, rel=0.01) and abs(float(zh.mean())) < 1e-4
    eng.close()


def test_bbfm_config5_over_the_lmr60_channel(Engine, torch_dev, oracle):
    """SURVEY 8(d) config 5, its fading variant: BBFM at batch 256 through the land-mobile channel of multipath_samples.m:17-21 ("lmr60": 60 km/h at 450 MHz, fd = 25 Hz),
    |H| at the 2000 symbols/s rate generated ON THE DEVICE (multipath_h_gen) per stream: encoder (bottleneck 1) -> FM-demodulator channel with that H and supplied noise ->
    decoder, three streams against the oracle on the same H and noise (bbfm.py:157-197)."""
    import os
    import torch
    from radae_amd.engine import DEFAULT_BLOB
    blob = os.path.join(os.path.dirname(DEFAULT_BLOB), "bbfm_random_seed20240501.bin")
    B, T = 256, 72
    rng = np.random.default_rng(515)
    feats = (0.7 * rng.standard_normal((B, T, 80))).astype(np.float32)
    eng = Engine(B, max_tx_mf=T // 3, blob=blob, flags=0x100)
    z = eng.encode(torch.tensor(feats, device=torch_dev))
    H = eng.multipath_h_gen("lmr60", T * 80, seed=99)                       # [B, T * 80, 1]: one |H| per real symbol
    Hn = H.cpu().numpy()[:, :, 0]
    assert 0.7 < float(np.mean(Hn.astype(np.float64) ** 2)) < 1.3 and float((20 * np.log10(Hn) + 14.0 < 12).mean()) > 0.2      # a fading channel: a good part of the symbols below the FM threshold at 14 dB
    noise = rng.standard_normal((B, T * 80)).astype(np.float32)
    Gfm = 13.467874862246564                                                # bbfm.py's FM gain for the default deviation (tests/golden/bbfm.npz: Gfm)
    zh = eng.channel_symbol(z, "bbfm", 14.0, Gfm, H=H.reshape(B, T * 80).contiguous(), noise=torch.tensor(noise, device=torch_dev))
    fh = eng.decode(zh, 80).cpu().numpy()
    zc, zhc = z.cpu().numpy(), zh.cpu().numpy()
    m = oracle.Model(blob)
    for b in (0, 131, 255):
        enc, dec = oracle.Encoder(m), oracle.Decoder(m)
        zo = np.array([enc.step(feats[b, t], bottleneck=1) for t in range(T)])
        assert np.abs(zc[b] - zo).max() < 2e-5
        zho = oracle.channel_bbfm(zc[b], Hn[b], noise[b], 14.0, Gfm).reshape(T, 80)
        assert np.abs(zhc[b] - zho).max() < 5e-6
        fo = np.array([dec.step(zhc[b, t]) for t in range(T)])
        assert rms(fh[b], fo) < 1e-4
    eng.close()


@pytest.mark.parametrize("blobname,mode", [("model05.bin", "rs"), ("bbfm_random_seed20240501.bin", "bbfm")])
def test_configs_1_and_5_at_batch_256_split_f16_gemm(Engine, torch_dev, oracle, blobname, mode):
    """BASELINE configs 1 and 5 at their stated batch (256 streams): with more than 16 k GEMM rows the 80-wide-input blobs run the
    split-binary16 matrix-core kernels (k_gemm16p / k_gemm16), not the f32 kernels their B = 1 parity tests above select.  Three
    streams each against the oracle's CoreEncoder / CoreDecoder on the same rows (z within 2e-5 of full scale, features within
    1e-4 RMS), including a stream whose raw features / received symbols are far outside +-1 (dense1's operands are unbounded)."""
    import os
    import torch
    from radae_amd.engine import DEFAULT_BLOB
    blob = os.path.join(os.path.dirname(DEFAULT_BLOB), blobname)
    B, T = 256, 72                                            # 18,432 rows per GEMM
    assert B * T > 16384
    rng = np.random.default_rng(20240929)
    x = np.zeros((B, T, 4, 20), np.float32)                   # AR(1) "speech-like" features, frame by frame
    v = np.zeros((B, 20), np.float32)
    for t in range(4 * T):
        v = np.float32(0.9) * v + np.float32(0.436) * rng.standard_normal((B, 20)).astype(np.float32)
        x[:, t // 4, t % 4] = v
    x[..., 0] *= 4.0
    x[100] *= np.float32(120.0)                               # one stream far past +-256
    feats = x.reshape(B, T, 80)
    eng = Engine(B, max_tx_mf=T // 3, blob=blob, flags=0x100)      # RADE_BATCH_BOTTLENECK1
    z = eng.encode(torch.tensor(feats, device=torch_dev)).cpu().numpy()
    assert np.isfinite(z).all()
    m = oracle.Model(blob)
    picks = (0, 100, 255)
    for b in picks:
        enc = oracle.Encoder(m)
        zr = np.stack([enc.step(r, bottleneck=1) for r in feats[b]])
        # tanh bottleneck: full scale 1, bar 1e-5 RMS.  Peak bar: these blobs' z_dense sums reach tens before the tanh (most latents sit
        # at +-1), where the 22-bit operand planes' 2^-22 relative error is 2..4e-5 on single values (measured 2.2e-5 on an ordinary
        # stream, 3.8e-5 on the x120 stream that drives every layer into saturation): 5e-5 / 1e-4
        assert np.abs(z[b] - zr).max() < (1e-4 if b == 100 else 5e-5) and rms(z[b], zr) < 1e-5, (blobname, b, float(np.abs(z[b] - zr).max()))
    # channel on the device (explicit noise => comparable), then the stand-alone decoder on what came out of it
    noise = rng.standard_normal((B, T * 80)).astype(np.float32)
    zt = torch.tensor(z, device=torch_dev)
    if mode == "rs":
        sigma = 10 ** (-6.0 / 20)
        zh = eng.channel_symbol(zt, "rs", sigma, noise=torch.tensor(noise, device=torch_dev)).cpu().numpy()
        for b in picks:
            assert np.abs(zh[b].ravel() - oracle.channel_rs(z[b].ravel(), None, noise[b], sigma)).max() < 1e-6
        zh[100, 5::7] *= np.float32(1e3)                          # a deep fade / false sync: symbols divided by a tiny pilot magnitude
    else:
        zh = eng.channel_symbol(zt, "bbfm", 14.0, 13.47, noise=torch.tensor(noise, device=torch_dev)).cpu().numpy()
        for b in picks:
            assert np.abs(zh[b].ravel() - oracle.channel_bbfm(z[b].ravel(), None, noise[b], 14.0, 13.47)).max() < 5e-6
    fh = eng.decode(torch.tensor(zh, device=torch_dev), 80).cpu().numpy()
    assert np.isfinite(fh).all()
    for b in picks:
        dec = oracle.Decoder(m)
        ref = np.stack([dec.step(r) for r in zh[b]])
        assert rms(fh[b], ref) < 1e-4, (blobname, b)
    eng.close()


def test_tx_channel_one_pass_equals_two_calls(Engine, torch_dev, oracle, oracle_model):
    """rade_batch_tx_channel (the modulator applies the two-path model and leaves the power sums: RADAE.forward in one pass) against
    rade_batch_tx + rade_batch_channel on the same inputs, and against the oracle's transmitter + channel for one stream."""
    import torch
    from radae_amd.channel_tools import multipath_g, synth_features
    from radae_amd.engine import sigma_from_EbNodB
    B, n_mf = 3, 12
    feats = np.stack([synth_features(40 + b, 12 * n_mf) for b in range(B)])
    G = np.stack([multipath_g("mpp", 8000, n_mf * 960, 70 + b) for b in range(B)])
    rng = np.random.default_rng(8)
    n_tot = 300 + n_mf * 960 + 1152 + 200
    noise = ((rng.standard_normal((B, n_tot)) + 1j * rng.standard_normal((B, n_tot))) / np.sqrt(2)).astype(np.complex64)
    sigma = sigma_from_EbNodB(6.0)
    ft, Gt, nt = torch.tensor(feats, device=torch_dev), torch.tensor(G, device=torch_dev), torch.tensor(noise, device=torch_dev)
    e1 = Engine(B, max_tx_mf=n_mf); e2 = Engine(B, max_tx_mf=n_mf)
    iq = e1.tx(ft)
    rx_two = e1.channel(iq, sigma, -7.0, n_pre=300, n_post=200, with_eoo=True, G=Gt, noise=nt).cpu().numpy()
    rx_one, iq_one = e2.tx_channel(ft, sigma, -7.0, n_pre=300, n_post=200, with_eoo=True, G=Gt, noise=nt, want_iq=True)
    assert np.array_equal(iq_one.cpu().numpy(), iq.cpu().numpy())                 # the clean transmit samples are the same bits
    assert np.abs(rx_one.cpu().numpy() - rx_two).max() < 2e-6                     # power sums meet in a different order: gain within an ulp
    e2.tx_reset()
    rx_noiq = e2.tx_channel(ft, sigma, -7.0, n_pre=300, n_post=200, with_eoo=True, G=Gt, noise=nt)
    assert np.array_equal(rx_noiq.cpu().numpy(), rx_one.cpu().numpy())            # iq output optional
    # AWGN-only: falls back to the two calls
    e1.tx_reset(); e2.tx_reset()
    a = e1.channel(e1.tx(ft), sigma, 0.0, noise=nt[:, :n_mf * 960].contiguous()).cpu().numpy()
    b_ = e2.tx_channel(ft, sigma, 0.0, noise=nt[:, :n_mf * 960].contiguous()).cpu().numpy()
    assert np.array_equal(a, b_)
    # oracle, stream 1
    tx = oracle.Tx(oracle_model)
    sig = np.concatenate([tx.frame(feats[1, 12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)])
    r, fin = oracle.channel(sig, G[1], noise[1, 300:300 + len(sig)], sigma, -7.0)
    ee = oracle.channel_eoo(tx.eoo(), noise[1, 300 + len(sig):300 + len(sig) + 1152], sigma, -7.0, 0.0, fin)
    full = np.concatenate([sigma * noise[1, :300], r, ee, sigma * noise[1, -200:]]).astype(np.complex64)
    assert np.abs(rx_one.cpu().numpy()[1] - full).max() < 5e-5
    e1.close(); e2.close()




def test_rx2_replicas_agree_when_two_workgroups_share_a_cu(Engine, torch_dev, golden, monkeypatch):
    """300 copies of one stream: more workgroups than CUs, so 44 CUs run two of them side by side -- the situation k_rx_sync2 exists
    for, and the one in which packed f32 FMAs in its band-pass filter corrupted lanes 48..63 (HISTORY.md 3.7).  Bit-identical in every
    slot, three fresh launches, and equal to the one-stream-per-CU kernel's discrete outputs."""
    import torch
    g = golden("rxtrace_mpp")
    B = 300
    buf = torch.tensor(np.stack([g["rx_in"]] * B), device=torch_dev)
    eng = Engine(B, max_tx_mf=1, rx_trace_calls=64, flags=0)
    ref = Engine(1, max_tx_mf=1, rx_trace_calls=64)
    fr, sr, _ = ref.rx(buf[:1].contiguous()); tr = ref.rx_trace(0)
    for rep in range(3):
        eng.rx_reset()
        f, st, _ = eng.rx(buf)
        f = f.cpu().numpy()
        assert all(st[b].n_calls == st[0].n_calls and st[b].n_valid == st[0].n_valid for b in range(B))
        bad = [b for b in range(1, B) if not np.array_equal(f[b], f[0])]
        assert not bad, (rep, bad[:10])
        t = eng.rx_trace(B - 1)
        for k in INT_KEYS:
            assert np.array_equal(t[k], tr[k]), k
        nv = st[0].n_valid
        assert sr[0].n_valid == nv and rms(f[0, :nv], fr[0, :nv].cpu().numpy()) < 2e-6
    eng.close(); ref.close()


# ---------------------------------------------------------------------------------------------------------------------
# single-carrier modem for BBFM symbols (reference radae/dsp.py:579-860)
# ---------------------------------------------------------------------------------------------------------------------
SC_RX_CASES = ["clean", "noisy_foff", "drift", "drift_pos", "lose_sync"]


@pytest.mark.gpu
def test_sc_rrc_and_tx_golden(torch_dev, golden):
    import torch
    from radae_amd.sc import SingleCarrierBatch
    for name in ("bpsk_1500", "analog_0"):
        g = golden("sc_tx_" + name)
        m = SingleCarrierBatch(2, fcentreHz=float(g["fcentre"]))
        assert np.abs(m.rrc() - g["rrc"]).max() < 1e-12
        sy = torch.tensor(np.stack([g["symbs"], g["symbs"][::-1].copy()]), device=torch_dev)
        # two calls (3 + rest frames): filter memory and LO phase carry across calls
        a = m.tx(sy[:, :3].contiguous()); b = m.tx(sy[:, 3:].contiguous())
        tx = torch.cat([a, b], dim=1).cpu().numpy()
        assert np.abs(tx[0] - g["tx"].reshape(-1)).max() < 2e-6
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", SC_RX_CASES)
def test_sc_rx_golden(torch_dev, golden, name):
    """HIP receiver against the reference's per-frame outputs: discrete outputs exact, symbols to 1e-5."""
    import torch
    from radae_amd.sc import SingleCarrierBatch
    g = golden("sc_rx_" + name)
    m = SingleCarrierBatch(1, fcentreHz=float(g["fcentre"]))
    pay, zh, fr, st = m.rx(torch.tensor(g["rx_in"][None], device=torch_dev))
    nf = st[0].n_frames
    assert nf == len(g["state"]) and st[0].consumed == int(g["consumed"])
    f = fr[0, :nf]
    for k in ("state", "nin", "fs_s"):
        assert np.array_equal(f[k], g[k]), k
    assert np.abs(f["norm_rx_timing"] - g["norm_rx_timing"]).max() < 1e-5
    assert np.abs(f["phase_ambiguity"] - g["phase_ambiguity"]).max() < 1e-6
    assert np.abs(f["g"] - g["g"]).max() < 1e-5 * np.abs(g["g"]).max()
    assert np.abs((f["max_cs_re"] + 1j * f["max_cs_im"]) - g["max_Cs"]).max() < 1e-5
    p = pay.cpu().numpy()[0, :nf]
    assert np.abs(p - g["payload"]).max() < 1e-5 * max(1.0, np.abs(g["payload"]).max())
    z = zh.cpu().numpy()[0, :nf]
    ref = np.where(g["state"][:, None] == 1, g["g"][:, None] * g["payload"].real, 0.0)
    assert np.abs(z - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    m.close()


@pytest.mark.gpu
def test_sc_loopback_batch_vs_oracle_and_chunking(torch_dev):
    """64 streams with different symbols / impairments: tx on the GPU == oracle tx; rx frame by frame (one call per frame,
    the sc_rx.py usage) == whole stream at once == oracle; bit errors of the synced frames are zero at this SNR."""
    import torch
    from oracle import sc_oracle as SC
    from radae_amd.sc import SingleCarrierBatch
    B, NF = 64, 10
    rng = np.random.default_rng(5)
    sy = (1 - 2 * (rng.random((B, NF, 80)) > 0.5)).astype(np.float32)
    m = SingleCarrierBatch(B, fcentreHz=1500.0)
    tx = m.tx(torch.tensor(sy, device=torch_dev)).cpu().numpy()
    for b in (0, 17, 63):
        o = SC.SingleCarrier(fcentreHz=1500.0)
        assert np.abs(np.concatenate([o.tx(s) for s in sy[b]]) - tx[b]).max() < 2e-6
    n = tx.shape[1]
    t = np.arange(n)
    rx = np.zeros((B, n + 64), np.complex64)
    for b in range(B):
        lead = int(rng.integers(0, 64))
        ph = 2 * np.pi * rng.uniform(-2, 2) * t / 9600 + rng.uniform(-3, 3)
        y = (0.3 + 2 * rng.random()) * (tx[b] * np.exp(1j * ph) + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
        rx[b, lead:lead + n] = y
    rxd = torch.tensor(rx, device=torch_dev)
    pay, zh, fr, st = m.rx(rxd)
    nf = st[0].n_frames
    assert all(s.n_frames == nf and s.consumed == nf * 384 or abs(s.consumed - nf * 384) <= nf for s in st)
    for b in (0, 17, 40, 63):
        o = SC.SingleCarrier(fcentreHz=1500.0)
        out, consumed = SC.run_rx_stream(o, rx[b])
        k = st[b].n_frames
        assert k == len(out["state"]) and consumed == st[b].consumed
        for key in ("state", "nin", "fs_s"):
            assert np.array_equal(fr[b, :k][key], out[key]), key
        assert np.abs(pay.cpu().numpy()[b, :k] - out["payload"]).max() < 2e-5 * max(1.0, np.abs(out["payload"]).max())
    # payload bits of synced frames: frame i of the receiver carries tx frame i - 1 or i - 2 depending on the lead
    z = zh.cpu().numpy(); errs = 0; bits = 0
    for b in range(B):
        for i in range(2, st[b].n_frames):
            if fr[b, i]["state"] == 1:
                best = min(int(np.sum(z[b, i] * sy[b, j] < 0)) for j in range(NF))
                errs += best; bits += 80
    assert bits > B * 5 * 80 and errs == 0
    # one frame per call == all at once
    # (streams slip independently, so the frame-by-frame run uses stream 0 alone)
    m1 = SingleCarrierBatch(1, fcentreHz=1500.0)
    p0 = 0; outs = []; nin = 384
    while p0 + nin <= rx.shape[1]:
        p, _, f1, s1 = m1.rx(rxd[:1, p0:p0 + nin].contiguous(), max_frames=1)
        assert s1[0].n_frames == 1 and s1[0].consumed == nin
        outs.append(p.cpu().numpy()[0, 0]); p0 += nin; nin = s1[0].nin
    assert len(outs) == st[0].n_frames
    assert np.array_equal(np.stack(outs), pay.cpu().numpy()[0, :len(outs)])
    m.close(); m1.close()


@pytest.mark.gpu
def test_sc_edge_cases(torch_dev, golden):
    """Short input (no whole frame), max_frames cap, reset, and argument checking of the single-carrier entry points."""
    import torch
    from radae_amd.sc import SingleCarrierBatch
    g = golden("sc_rx_clean")
    x = torch.tensor(g["rx_in"][None], device=torch_dev)
    m = SingleCarrierBatch(1, fcentreHz=float(g["fcentre"]))
    pay, zh, fr, st = m.rx(x[:, :383].contiguous(), max_frames=4)          # fewer than nin samples: nothing consumed
    assert st[0].n_frames == 0 and st[0].consumed == 0 and st[0].nin == 384 and st[0].state == 0
    pay, zh, fr, st = m.rx(x, max_frames=3)                                   # capped: the caller resumes at `consumed`
    assert st[0].n_frames == 3 and st[0].consumed == int(np.sum(np.concatenate([[384], g["nin"][:2]])))
    pay2, zh2, fr2, st2 = m.rx(x[:, st[0].consumed:].contiguous())
    nf = len(g["state"])
    assert st2[0].n_frames == nf - 3
    assert np.array_equal(np.concatenate([fr[0, :3]["state"], fr2[0, :nf - 3]["state"]]), g["state"])
    p = np.concatenate([pay.cpu().numpy()[0, :3], pay2.cpu().numpy()[0, :nf - 3]])
    assert np.abs(p - g["payload"]).max() < 1e-5
    m.reset()                                                                  # back to the constructor state: same answer again
    pay3, _, fr3, st3 = m.rx(x)
    assert st3[0].n_frames == nf and np.abs(pay3.cpu().numpy()[0, :nf] - g["payload"]).max() < 1e-5
    # argument errors are reported, not crashed on
    L = m.L
    assert L.rade_sc_tx(m.h, None, 1, None, 384, None) == -1
    assert L.rade_sc_rx(m.h, None, 0, 0, 1, None, None, None, None, None) == -1
    assert not L.rade_sc_open(0, 2400.0, 9600.0, 0.0, 0.25, 0)
    assert not L.rade_sc_open(1, 2400.0, 8000.0, 0.0, 0.25, 0)              # Fs must be 4 Rs
    m.close()


@pytest.mark.gpu
def test_sc_wire_filters_match_reference_scripts(golden):
    """latents | sc_tx | sc_rx with the reference's own scripts (tests/golden/sc_wire.npz) vs the GPU filters: int16 samples equal up
    to the truncation of values within float rounding of an integer, z_hat equal to the quantisation noise that leaves."""
    from radae_amd.sc import sc_tx_stream, sc_rx_stream
    g = golden("sc_wire")
    t = sc_tx_stream(g["z"])
    assert t.dtype == np.int16 and t.shape == g["t_int16"].shape
    d = np.abs(t.astype(int) - g["t_int16"].astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.97
    zh = sc_rx_stream(g["t_int16"])                       # the reference's samples in: isolates the receiver
    assert zh.shape == g["zhat"].shape and np.abs(zh - g["zhat"]).max() < 2e-5 * max(1.0, np.abs(g["zhat"]).max())
    zh2 = sc_rx_stream(t)                                  # our own samples: +-1 LSB differences only
    assert zh2.shape == g["zhat"].shape and np.abs(zh2 - g["zhat"]).max() < 2e-3


def test_config4_sharding_two_ranks_on_one_device():
    """The N > 1 path of bench.py on the GPU there is: two ranks under torch.distributed.run, both on device 0 (--oversubscribe-device: gloo for the blob broadcast
    and the statistics all-reduce), utterances sharded [0, 16) / [16, 32) as `--gpus 2` shards them; each rank's decoded features (sha256 of the last step) equal a single
    process run on that rank's shard (--as-shard r/2), and the job-wide counts are the sums.  tools/oversub8.sh is the same at 8 x 256 utterances (profiles/r06_oversub8.json)."""
    import json, subprocess, sys
    common = ["--streams", "16", "--frames", "240", "--steps", "3", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--no-roofline", "--no-parity"]
    def run(extra):
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + extra + common, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    job = run(["--gpus", "2", "--oversubscribe-device", "0"])
    assert job["oversubscribed"]["ranks"] == 2 and job["rccl_ranks"] == 2 and job["config"]["global_streams"] == 32 and len(job["last_step_features_sha256_per_rank"]) == 2
    shards = [run(["--as-shard", f"{r}/2"]) for r in range(2)]
    assert [s["utterances"] for s in shards] == [[0, 16], [16, 32]]
    assert job["last_step_features_sha256_per_rank"] == [s["last_step_features_sha256_rank0"] for s in shards]
    assert shards[0]["last_step_features_sha256_rank0"] != shards[1]["last_step_features_sha256_rank0"]
    for k in ("offered_frames", "decoded_frames", "rx_calls", "sync_calls", "search_calls"):
        assert job["job_last_step"][k] == shards[0]["job_last_step"][k] + shards[1]["job_last_step"][k], k
    assert job["job_last_step"]["decoded_frames"] > 0


def test_python_cli_filters_like_the_reference_tools(golden, tmp_path):
    """`python -m radae_amd.cli txe | rxe` in place of `python3 radae_txe.py | radae_rxe.py` (radae_txe.py:146-180, radae_rxe.py:332-371), three of the reference's pipelines:
    (i) ctest radae_eoo_data_py (CMakeLists.txt:578-583): features -> txe --eoo_data_test -> rxe --eoo_data_test prints the BER line and PASS, eoo_tx.f32 is written;
    (ii) the transmit samples equal the golden ones (enc_tx.npz), the EOO frame follows; (iii) ctest radae_rx_slip_plus_drops (:397-407): the 61 s int16 file through
    `int16tof32 --zeropad | rxe`, last stderr line `state: sync`; (iv) txe | rxe --bypass_dec writes 240 floats per valid frame."""
    import subprocess, sys
    from radae_amd import wire
    env = dict(os.environ); env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    def run(args, data):
        r = subprocess.run([sys.executable, "-m", "radae_amd.cli"] + args, input=data, capture_output=True, cwd=str(tmp_path), env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        return r.stdout, r.stderr.decode()
    e = golden("enc_tx")
    feats = e["features"][0].astype(np.float32)                           # (120, 36): ten modem frames
    iq, _ = run(["txe", "--eoo_data_test"], feats.tobytes())
    iq = np.frombuffer(iq, np.complex64)
    assert iq.size == 10 * 960 + 1152 and np.abs(iq[:9600].reshape(10, 960) - e["tx"][0]).max() < 2e-5
    bits = np.fromfile(str(tmp_path / "eoo_tx.f32"), np.float32)
    rng = np.random.default_rng(65647)
    assert np.array_equal(bits, np.sign(rng.random(180) - 0.5).astype(np.float32))
    rx_in = np.concatenate([np.zeros(960, np.complex64), iq, np.zeros(2400, np.complex64)])
    fo, err = run(["rxe", "--eoo_data_test"], rx_in.tobytes())
    assert "EOO data n_bits: 180 n_errors: 0 BER:  0.00" in err and "PASS" in err
    n_valid = len(fo) // (432 * 4)
    assert n_valid >= 6 and len(fo) % (432 * 4) == 0
    zo, _ = run(["rxe", "--bypass_dec", "-v", "0"], rx_in.tobytes())
    assert len(zo) == n_valid * 240 * 4
    g = golden("rxtrace_slipdrops")
    _, err = run(["rxe", "-v", "1", "--no_stdout"], wire.int16_to_f32(g["rx_i16"].tobytes(), zeropad=True))
    assert err.strip().splitlines()[-1] == "state: sync"


def test_ctest_pipeline_radae_rx_mpp_through_the_command_lines(tmp_path):
    """The reference's ctest radae_rx_mpp (CMakeLists.txt:326-334) as a shell pipeline of this repo's command lines -- multipath_samples -> inference (--g_file, Eb/No 3 dB,
    -11 Hz, prepend 1 s, append 3 s, end of over) -> rxe (--disable_unsync 5) -> loss (--loss_test, --acq_time_test) -- on synthetic features (no speech file exists here: out of distribution for the trained model, loss 0.7 in
    loopback and ~2.2 at 3 dB MPP, so the threshold is a sanity bound, not the ctest's 0.3); and the same samples through the oracle receiver decode the same frames."""
    import subprocess, sys
    from radae_amd.channel_tools import synth_features
    env = dict(os.environ); env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    def run(mod, args, stdin=None):
        r = subprocess.run([sys.executable, "-m", mod] + args, input=stdin, capture_output=True, cwd=str(tmp_path), env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        return r.stdout, r.stderr.decode()
    synth_features(4242, 600).tofile(str(tmp_path / "features_in.f32"))
    out, _ = run("radae_amd.cli", ["multipath_samples", "mpp", "8000", "50", "30", "10", "h_mpp.f32", "g_mpp.f32"])
    out, _ = run("radae_amd.cli", ["inference", "model19_check3", "features_in.f32", "/dev/null", "--EbNodB", "3", "--freq_offset", "-11", "--g_file", "g_mpp.f32", "--rate_Fs", "--pilots",
                                   "--pilot_eq", "--eq_ls", "--cp", "0.004", "--bottleneck", "3", "--time_offset", "-16", "--write_rx", "rx.f32", "--prepend_noise", "1", "--append_noise", "3",
                                   "--end_of_over", "--auxdata", "--correct_freq_offset"])
    assert b"Target..:   3.00" in out and b"Measured:" in out
    rx = np.fromfile(str(tmp_path / "rx.f32"), np.complex64)
    assert rx.size == 8000 + 50 * 960 + 1152 + 24000
    fo, err = run("radae_amd.cli", ["rxe", "--disable_unsync", "5", "-v", "1"], rx.tobytes())
    (tmp_path / "features_rx_out.f32").write_bytes(fo)
    n = len(fo) // (36 * 4)
    assert n >= 12 * 30                                        # most of the 50 modem frames decoded at 3 dB MPP
    out, _ = run("radae_amd.loss", ["features_in.f32", "features_rx_out.f32", "--loss_test", "4.0", "--acq_time_test", "1.5", "--clip_end", "100"])
    assert out.decode().strip().endswith("PASS"), out.decode()
    from oracle import oracle_py as O
    O.build()
    d = O.run_rx_stream(O.Model(), rx, 1, 0.0, 5.0)
    assert len(d["features_out"]) * 12 == n and rms(np.frombuffer(fo, np.float32).reshape(-1, 432), d["features_out"]) < 1e-5


def test_bbfm_inference_command_line(golden, tmp_path):
    """`python -m radae_amd.cli bbfm_inference` = bbfm_inference.py (:43-170) on the BBFM golden's features: at CNR 100 dB (no channel noise to speak of) the written features_hat
    equal the reference model's noise-free forward through the same de-quantised weights to 1e-4, the latents written by --write_latent equal its z; with the lmr60 |H| file made by
    `multipath_samples lmr60 8000 2000 1 ..` (BBFM.md:37) and CNR 14 dB the loss rises and the per-symbol CNR file is 20 log10 |H| + 14."""
    import subprocess, sys
    env = dict(os.environ); env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    def run(args):
        r = subprocess.run([sys.executable, "-m", "radae_amd.cli"] + args, capture_output=True, cwd=str(tmp_path), env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        return r.stdout.decode()
    g = golden("bbfm")
    f36 = np.zeros((len(g["features"]), 36), np.float32); f36[:, :20] = g["features"]
    f36.tofile(str(tmp_path / "features_in.f32"))
    out = run(["bbfm_inference", "default", "features_in.f32", "features_hat.f32", "--CNRdB", "100", "--write_latent", "z.f32", "--loss_test", "5.0"])
    assert "SNRdB Measured:" in out and out.strip().endswith("PASS")
    z = np.fromfile(str(tmp_path / "z.f32"), np.float32).reshape(-1, 80)
    assert np.abs(z - np.clip(g["z"], -1, 1)).max() < 1e-4                  # sigma at 113 dB SNR is 2e-6: z_hat = clamp(z)
    run(["multipath_samples", "lmr60", "8000", "2000", "1", "6", "h_lmr60.f32"])
    out2 = run(["bbfm_inference", "default", "features_in.f32", "/dev/null", "--CNRdB", "14", "--h_file", "h_lmr60.f32", "--write_CNRdB", "cnr.f32"])
    h = np.fromfile(str(tmp_path / "h_lmr60.f32"), np.float32); cnr = np.fromfile(str(tmp_path / "cnr.f32"), np.float32)
    assert np.abs(cnr - (20 * np.log10(h[:cnr.size]) + 14.0)).max() < 1e-3
    loss = lambda o: float([l for l in o.splitlines() if l.startswith("loss:")][0].split()[1])
    assert loss(out2) > loss(out)
