"""SURVEY.md section 8 row a17 -- the DNNw weight blob, pinned against the reference's OWN exporter.

The fixtures (tests/golden/dnnw_export_A.bin, dnnw_export.npz, weights_check.npz) are written by oracle/gen_golden_dnnw.py, which runs
`/root/reference/export_rade_weights.py: c_export` (:54-172 -> wexchange/c_export/common.py) on two seeded synthetic checkpoints and packs the emitted C
arrays as `/root/reference/src/write_rade_weights.c:51-74` does.  Checkpoint A is "lossless" (oracle/dnnw_synth.py): the exporter's quantisation returns its
weights exactly, so the ground truth every reader must reproduce BIT FOR BIT is the checkpoint itself -- regenerated here from its seed, no reader of this
repo is involved in making it.  The GRU input matrices are genuinely sparse (a third of the 8x4 blocks absent, a trailing input block column absent
everywhere).  Checkpoint B (Gaussian floats) is the byte-identity target of the writer."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from oracle import dnnw_synth
from radae_amd import dnnw

HERE = os.path.dirname(os.path.abspath(__file__))
BLOB_A = os.path.join(HERE, "golden", "dnnw_export_A.bin")


@pytest.fixture(scope="module")
def fx(golden):
    return golden("dnnw_export")


@pytest.fixture(scope="module")
def ckpt_a(fx):
    return dnnw_synth.synth_model(int(fx["seeds"][0]), lossless=True)


def test_fixture_blob_is_the_exported_one(fx):
    blob = open(BLOB_A, "rb").read()
    assert hashlib.sha256(blob).hexdigest() == str(fx["A_sha256"])
    rec = dnnw.read_records(BLOB_A)
    assert list(rec.keys()) == list(fx["A_record_names"]) and len(rec) == 158            # same record list and order as bin/model19_check3.bin
    m19 = dnnw.read_records(os.path.join(os.path.dirname(HERE), "weights", "model19_check3.bin"))
    assert list(m19.keys()) == list(rec.keys())
    assert all(m19[k].dtype == rec[k].dtype for k in rec)
    # dense-in-sparse-clothing in model19, genuinely sparse here
    for k in rec:
        if k.endswith("_weights_idx"):
            assert rec[k].size < m19[k].size, k


def test_python_reader_returns_the_checkpoint_bit_for_bit(ckpt_a):
    got, want = dnnw_synth.tensors(dnnw.load_model(BLOB_A)), dnnw_synth.tensors(ckpt_a)
    for k in want:
        assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), k


def test_reader_agrees_with_the_exporters_own_q_scale_subias(fx, ckpt_a):
    """The exporter's records (captured around `print_linear_layer`): scale as `compute_scaling` returned it, q as `quantize_weight` returned it (digest, in
    the exporter's (n_in, n_out) orientation with gates z,r,n and conv rows k * n_in + i), subias.  The reader's output, taken back to that orientation,
    must be q * scale exactly."""
    rec = dnnw.read_records(BLOB_A)
    m = dnnw.load_model(BLOB_A)
    swap = dnnw._swap_gates
    layers = {}
    for i in range(5):
        for side, grus, convs in (("enc", m.enc_gru, m.enc_conv), ("dec", m.dec_gru, m.dec_conv)):
            layers[f"{side}_gru{i+1}_input"] = swap(grus[i].w_ih).T
            layers[f"{side}_gru{i+1}_recurrent"] = swap(grus[i].w_hh).T
            c = convs[i].w
            layers[f"{side}_conv{i+1}"] = c.transpose(2, 1, 0).reshape(-1, c.shape[0])
        layers[f"dec_glu{i+1}"] = m.dec_glu[i].w.T
    assert len(layers) == 35
    for name, w_io in layers.items():
        scale = fx[f"A/{name}/scale"]
        q = w_io / scale[None, :]
        assert np.array_equal(q, np.round(q)) and np.abs(q).max() == 127, name
        assert hashlib.sha256(np.ascontiguousarray(q).astype(np.int8).tobytes()).hexdigest() == str(fx[f"A/{name}/q_sha256"]), name
        assert np.array_equal(q[:8, :16].astype(np.int8), fx[f"A/{name}/q_head"]), name
        assert np.array_equal(rec[name + "_scale"], (scale / np.float32(127)).astype(np.float32)), name
        assert np.array_equal(rec[name + "_subias"], fx[f"A/{name}/subias"].astype(np.float32)), name
        if f"A/{name}/kept_blocks" in fx.files:
            n_in, n_out = w_io.shape
            kept = np.abs(w_io).reshape(n_in // 4, 4, n_out // 8, 8).sum(axis=(1, 3)).T > 0
            assert np.array_equal(np.packbits(kept), fx[f"A/{name}/kept_blocks"]) and kept.sum() == int(fx[f"A/{name}/n_kept"]) < kept.size
            assert rec[name + "_weights_idx"].size == n_out // 8 + kept.sum() and rec[name + "_weights_int8"].size == 32 * kept.sum()


def test_oracle_reader_returns_the_checkpoint_bit_for_bit(oracle, golden, ckpt_a):
    """orc_model_load on the exported blob == the checkpoint; weights_check.npz (statistics of the CHECKPOINT's tensors, written by gen_golden_dnnw.py, not by
    any reader) is the travelling form of the same statement."""
    om = oracle.Model(BLOB_A)
    want = dnnw_synth.tensors(ckpt_a)
    w = golden("weights_check")
    assert len(w.files) == len(want)
    for k, v in want.items():
        a = np.ascontiguousarray(v.transpose(0, 2, 1) if v.ndim == 3 else v).ravel()          # conv: [out][tap][in]
        t = om.tensor(k)
        assert np.array_equal(t, a), k
        t64 = t.astype(np.float64)
        assert np.allclose(np.array([t64.size, t64.sum(), np.abs(t64).sum()] + list(t64[:8]) + list(t64[-4:])), w[k], rtol=1e-12, atol=0), k


def test_oracle_reader_equals_python_reader_on_model19(oracle_model):
    """consistency on the shipped blob (no exporter truth exists for it: the checkpoint is not in the reference tree)"""
    from radae_amd import engine
    for k, v in dnnw_synth.tensors(dnnw.load_model(engine.DEFAULT_BLOB)).items():
        a = np.ascontiguousarray(v.transpose(0, 2, 1) if v.ndim == 3 else v).ravel()
        assert np.array_equal(oracle_model.tensor(k), a), k


def test_host_reader_returns_the_checkpoint_bit_for_bit(ckpt_a):
    """rd_model_parse (radae_amd/csrc/rade_host.c), the reader the engine uploads from."""
    from radae_amd import engine
    from test_host_cpu import Model
    lib = engine.load_library()
    blob = open(BLOB_A, "rb").read()
    m = Model()
    lib.rd_model_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Model)]
    assert lib.rd_model_parse(blob, len(blob), C.byref(m)) == 0
    arr = lambda p, n: np.ctypeslib.as_array(p, shape=(n,))
    a = ckpt_a
    for lin, ref in ((m.enc_dense1, a.enc_dense1), (m.enc_zdense, a.enc_zdense), (m.dec_dense1, a.dec_dense1), (m.dec_output, a.dec_output)):
        assert np.array_equal(arr(lin.w, ref.w.size).reshape(ref.w.shape), ref.w) and np.array_equal(arr(lin.b, ref.b.size), ref.b)
    for i in range(5):
        for g, r in ((m.enc_gru[i], a.enc_gru[i]), (m.dec_gru[i], a.dec_gru[i])):
            assert (g.n_in, g.hid) == (r.w_ih.shape[1], r.w_hh.shape[1])
            assert np.array_equal(arr(g.w_ih, r.w_ih.size).reshape(r.w_ih.shape), r.w_ih) and np.array_equal(arr(g.w_hh, r.w_hh.size).reshape(r.w_hh.shape), r.w_hh)
            assert np.array_equal(arr(g.b_ih, r.b_ih.size), r.b_ih) and np.array_equal(arr(g.b_hh, r.b_hh.size), r.b_hh)
            # row scales follow their rows through the gate swap: every row is an integer multiple of its scale, with |q| = 127 somewhere
            s = arr(g.s_ih, r.w_ih.shape[0])
            q = r.w_ih / s[:, None]
            assert np.array_equal(q, np.round(q)) and np.array_equal(np.abs(q).max(axis=1), np.full(len(s), 127.0))
        for c, r in ((m.enc_conv[i], a.enc_conv[i]), (m.dec_conv[i], a.dec_conv[i])):
            cw = r.w.transpose(0, 2, 1).reshape(r.w.shape[0], -1)                             # [out][tap][in]
            assert np.array_equal(arr(c.w, cw.size).reshape(cw.shape), cw) and np.array_equal(arr(c.b, r.b.size), r.b)
        assert np.array_equal(arr(m.dec_glu[i].w, 96 * 96).reshape(96, 96), a.dec_glu[i].w) and not arr(m.dec_glu[i].b, 96).any()
    lib.rd_model_free.argtypes = [C.POINTER(Model)]; lib.rd_model_free(C.byref(m))
    # a block position outside the layer's inputs is rejected (n_in is the architecture's, not inferred)
    rec_off = blob.index(b"enc_gru1_input_weights_idx") - 20
    bad = bytearray(blob); bad[rec_off + 64 + 4:rec_off + 64 + 8] = (64).to_bytes(4, "little")     # first block position -> 64 (layer has 64 inputs)
    assert lib.rd_model_parse(bytes(bad), len(bad), C.byref(Model())) != 0


def test_writer_is_byte_identical_to_the_reference_export(fx, ckpt_a, tmp_path):
    """dnnw.write_blob (what manufactures the BBFM blob) against export_rade_weights.py + the C compiler + write_rade_weights.c: the same bytes, for the
    lossless checkpoint and for Gaussian floats (float32 scale arithmetic, round-half-even quantiser, float64 subias sums in the exporter's memory order)."""
    p = str(tmp_path / "a.bin")
    dnnw.write_blob(ckpt_a, p)
    assert open(p, "rb").read() == open(BLOB_A, "rb").read()
    p = str(tmp_path / "b.bin")
    dnnw.write_blob(dnnw_synth.synth_model(int(fx["seeds"][1]), lossless=False), p)
    rec = dnnw.read_records(p)
    assert list(rec.keys()) == list(fx["B_record_names"])
    bad = [n for n, h in zip(fx["B_record_names"], fx["B_record_sha256"]) if hashlib.sha256(rec[str(n)].tobytes()).hexdigest()[:16] != str(h)]
    assert not bad, bad
    assert hashlib.sha256(open(p, "rb").read()).hexdigest() == str(fx["B_sha256"])
    for name in ("enc_gru3_input", "dec_conv2", "dec_glu4", "dec_gru1_recurrent"):
        assert np.array_equal(rec[name + "_scale"], (fx[f"B/{name}/scale"] / np.float32(127)).astype(np.float32))
        assert np.array_equal(rec[name + "_subias"], fx[f"B/{name}/subias"].astype(np.float32))
    # the exporter's failure mode is kept: an all-zero output column has scale 0 (common.py:132-137 raises on the NaNs)
    z = dnnw_synth.synth_model(3, lossless=False); z.dec_glu[0].w[5] = 0
    with pytest.raises(ValueError):
        dnnw.write_blob(z, str(tmp_path / "z.bin"))


def test_oracle_core_on_the_exported_blob_vs_reference_modules(oracle, fx):
    """The CPU oracle's core encoder / decoder on the reference-exported blob against the reference's float modules holding the same checkpoint."""
    m = oracle.Model(BLOB_A)
    e, e1, d = oracle.Encoder(m), oracle.Encoder(m), oracle.Decoder(m)
    z = np.stack([e.step(f) for f in fx["A/run/features"]]); z1 = np.stack([e1.step(f, bottleneck=1) for f in fx["A/run/features"]])
    fo = np.stack([d.step(v) for v in fx["A/run/z_hat"]])
    for got, want in ((z, fx["A/run/z"]), (z1, fx["A/run/z_bottleneck1"]), (fo, fx["A/run/features_out"])):
        assert np.sqrt(np.mean((got - want) ** 2)) < 2e-6 and np.abs(got - want).max() < 2e-5


# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_core_encoder_decoder_on_the_exported_blob_vs_reference_modules(fx):
    """rade_core_encoder / rade_core_decoder (include/rade_core.h) and the batched rade_batch_encode / rade_batch_decode, loaded from the blob the REFERENCE's
    exporter wrote, against the reference's float modules holding the same (lossless) checkpoint, one 40 ms step per call (fixture from
    oracle/gen_golden_dnnw.py step 5).  Tolerance: north_star's 1e-4 RMS; measured ~1e-6."""
    import torch
    from radae_amd.core import CoreDecoder, CoreEncoder
    from radae_amd.engine import BatchEngine
    feats, z_ref, z_hat, f_ref = fx["A/run/features"], fx["A/run/z"], fx["A/run/z_hat"], fx["A/run/features_out"]
    rms = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
    enc, dec = CoreEncoder(BLOB_A, 84), CoreDecoder(BLOB_A, 84)
    z = np.stack([enc.step(f) for f in feats]); fo = np.stack([dec.step(v) for v in z_hat])
    enc.reset()
    z1 = np.stack([enc.step(f, bottleneck=1) for f in feats])                              # RADAE's default bottleneck: z = tanh(.) (rade_enc.c:110-113)
    enc.close(); dec.close()
    assert rms(z1, fx["A/run/z_bottleneck1"]) < 1e-5
    assert rms(z, z_ref) < 1e-5 and np.abs(z - z_ref).max() < 1e-4, (rms(z, z_ref), np.abs(z - z_ref).max())
    assert rms(fo, f_ref) < 1e-5 and np.abs(fo - f_ref).max() < 1e-4, (rms(fo, f_ref), np.abs(fo - f_ref).max())
    assert rms(z, 0 * z) > 0.1 and rms(fo, 0 * fo) > 0.1
    dev = torch.device("cuda:0")
    eng = BatchEngine(2, max_tx_mf=8, blob=BLOB_A)
    zb = eng.encode(torch.tensor(np.stack([feats, feats[::-1].copy()]), device=dev)).cpu().numpy()
    fb = eng.decode(torch.tensor(np.stack([z_hat, z_hat]), device=dev), 84).cpu().numpy()
    eng.close()
    assert rms(zb[0], z_ref) < 1e-5 and rms(fb[0], f_ref) < 1e-5 and rms(fb[1], f_ref) < 1e-5
