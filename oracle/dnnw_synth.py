"""Seeded synthetic RADE core weights for pinning the DNNw blob reader / writer against the reference's own exporter.

TEST INFRASTRUCTURE -- imported by oracle/gen_golden_dnnw.py (build container, where the reference's exporter runs on these weights)
and by tests/ (which regenerate the same weights from the seed instead of carrying 7 MB of floats).  numpy only: `np.random.Generator(PCG64)`
streams are stable across numpy versions.

Two flavours of one architecture (radae_base.py:239-251 encoder, :377-393 decoder; torch orientation, torch gate order r,z,n):
  * lossless=True: every int8-exported layer holds w[o][i] = q[o][i] * 2^-k(o) with integer |q| <= 127, one |q| = 127 per output row and
    |q_even + q_odd| <= 129 for every input pair, so that the exporter's scale rule (wexchange/c_export/common.py:180-194) returns exactly
    2^-k(o) and its quantiser (:132-137) returns exactly q: the blob then holds the checkpoint's weights WITHOUT loss, and a correct reader
    must give them back bit for bit.  k(o) varies from row to row, which pins the row <-> scale association through the exporter's gate swap.
  * lossless=False: Gaussian float weights; exercises the exporter's float32 scale arithmetic and rounding (for the writer's byte-identity test).
GRU input matrices are genuinely sparse in both: about a third of the 4-input x 8-output blocks of the exported matrix are zero, every 8-output group keeps at least one block (an all-zero output makes the exporter divide 0 / 0), and the LAST input block column of enc_gru3 / dec_gru3 is empty in every group (so that a reader cannot infer n_in from the index list).
"""
from __future__ import annotations

import numpy as np

from radae_amd import dnnw

ENC_GRU_IN = (64, 224, 384, 544, 704)
ENC_CONV_IN = (128, 288, 448, 608, 768)
DEC_GRU_IN = (96, 224, 352, 480, 608)
DEC_CONV_IN = (192, 320, 448, 576, 704)


def _block_mask(rng, n_out, n_in, layer_tag):
    """keep[o // 8][i // 4] of the EXPORTED matrix.  The exporter swaps gates r <-> z in whole blocks of n_out / 3 rows (a multiple of 8), so an
    8-row group of the torch matrix is an 8-column group of the exported one."""
    keep = rng.random((n_out // 8, n_in // 4)) > 0.33
    for g in range(n_out // 8):
        if not keep[g].any():
            keep[g, rng.integers(n_in // 4)] = True
    if layer_tag == 3:
        keep[:, -1] = False                     # last input block column unused everywhere
    return keep


def _int8_rows(rng, n_out, n_flat, lossless, keep=None, fan=None):
    """(n_out, n_flat) matrix in the exporter's flat input order (pairs = consecutive flat inputs)."""
    if lossless:
        q = rng.integers(-64, 65, size=(n_out, n_flat)).astype(np.int64)
        k0 = int(np.round(np.log2(37.0 * np.sqrt(fan or n_flat) * 0.6)))
        kk = k0 + rng.integers(0, 3, size=n_out)
        if keep is not None:
            q *= np.repeat(np.repeat(keep, 8, axis=0), 4, axis=1)
        for o in range(n_out):
            if keep is not None:
                cols = np.flatnonzero(keep[o // 8])
                j = 4 * int(cols[rng.integers(cols.size)]) + 2 * int(rng.integers(2))
            else:
                j = 2 * int(rng.integers(n_flat // 2))
            q[o, j] = 127 * (1 if rng.random() < 0.5 else -1)
            q[o, j + 1] = -np.sign(q[o, j]) * int(rng.integers(0, 65))      # |pair sum| <= 127
        return (q.astype(np.float32) * np.exp2(-kk.astype(np.float32))[:, None]).astype(np.float32)
    w = (rng.standard_normal((n_out, n_flat)) / np.sqrt(fan or n_flat) * 1.5).astype(np.float32)
    if keep is not None:
        w *= np.repeat(np.repeat(keep, 8, axis=0), 4, axis=1).astype(np.float32)
    return w


def _floats(rng, n_out, n_in):
    return (rng.standard_normal((n_out, n_in)) / np.sqrt(n_in)).astype(np.float32), (0.1 * rng.standard_normal(n_out)).astype(np.float32)


def _bias(rng, n):
    return (0.1 * rng.standard_normal(n)).astype(np.float32)


def _gru(rng, n_in, H, lossless, tag):
    keep = _block_mask(rng, 3 * H, n_in, tag)
    return dnnw.GRU(_int8_rows(rng, 3 * H, n_in, lossless, keep=keep), _int8_rows(rng, 3 * H, H, lossless, fan=4 * H), _bias(rng, 3 * H), _bias(rng, 3 * H))


def _conv(rng, n_in, n_out, dil, lossless):
    flat = _int8_rows(rng, n_out, 2 * n_in, lossless)                   # exporter row index = k * n_in + i (common.py:307-311)
    return dnnw.Conv(np.ascontiguousarray(flat.reshape(n_out, 2, n_in).transpose(0, 2, 1)), _bias(rng, n_out), dil)


def synth_model(seed: int, lossless: bool) -> dnnw.Model:
    rng = np.random.Generator(np.random.PCG64(seed))
    return dnnw.Model(
        enc_dense1=dnnw.Dense(*_floats(rng, 64, 84)),
        enc_gru=[_gru(rng, ENC_GRU_IN[i], 64, lossless, i + 1) for i in range(5)],
        enc_conv=[_conv(rng, ENC_CONV_IN[i], 96, dnnw.ENC_DILATION[i], lossless) for i in range(5)],
        enc_zdense=dnnw.Dense(*_floats(rng, 80, 864)),
        dec_dense1=dnnw.Dense(*_floats(rng, 96, 80)),
        dec_gru=[_gru(rng, DEC_GRU_IN[i], 96, lossless, i + 1) for i in range(5)],
        dec_glu=[dnnw.Dense(_int8_rows(rng, 96, 96, lossless), np.zeros(96, np.float32)) for _ in range(5)],
        dec_conv=[_conv(rng, DEC_CONV_IN[i], 32, 1, lossless) for i in range(5)],
        dec_output=dnnw.Dense(*_floats(rng, 84, 736)),
    )


def tensors(m: dnnw.Model):
    """name -> array, torch orientation (conv as [out][in][k])."""
    d = {"enc_dense1_w": m.enc_dense1.w, "enc_dense1_b": m.enc_dense1.b, "enc_zdense_w": m.enc_zdense.w, "enc_zdense_b": m.enc_zdense.b,
         "dec_dense1_w": m.dec_dense1.w, "dec_dense1_b": m.dec_dense1.b, "dec_output_w": m.dec_output.w, "dec_output_b": m.dec_output.b}
    for i in range(5):
        for side, grus, convs in (("enc", m.enc_gru, m.enc_conv), ("dec", m.dec_gru, m.dec_conv)):
            g, c = grus[i], convs[i]
            d[f"{side}_gru{i+1}_w_ih"] = g.w_ih; d[f"{side}_gru{i+1}_w_hh"] = g.w_hh
            d[f"{side}_gru{i+1}_b_ih"] = g.b_ih; d[f"{side}_gru{i+1}_b_hh"] = g.b_hh
            d[f"{side}_conv{i+1}_w"] = c.w; d[f"{side}_conv{i+1}_b"] = c.b
        d[f"dec_glu{i+1}_w"] = m.dec_glu[i].w
    return d
