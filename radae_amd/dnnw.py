"""DNNw weight-blob reader (host side, numpy only).

The RADAE reference ships its deployed model as an Opus-style "DNNw" blob
(`bin/model19_check3.bin`); the layout is produced by
`/root/reference/src/write_rade_weights.c:51-74` from tables emitted by
`weight-exchange/wexchange/c_export/common.py` (dense :290-293, int8 8x4 blocks :59-69,
"sparse" GRU input blocks :140-176, scale/subias :263-271, GRU gate swap :360-368,
conv tap flattening :307-311).

This module inverts that export into plain fp32 matrices in *row-major [out][in]* form
(the orientation torch.nn.Linear uses) so that the same numbers can be
  * loaded into the C/HIP engine (it has its own C twin of this reader, `csrc/rade_host.c:rd_model_parse`),
  * loaded into the reference's PyTorch modules by `oracle/gen_golden.py`.

De-quantisation rule: the int8 value q and the per-output `scale` array satisfy
w = q * scale * 127 (the exporter stores scale/127, common.py:267).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Dict

import numpy as np

_TYPE_F32, _TYPE_I32, _TYPE_I8 = 0, 1, 3
_HDR = 64


def read_records(path: str) -> Dict[str, np.ndarray]:
    """Parse every record of a DNNw blob into a flat numpy array keyed by name."""
    blob = open(path, "rb").read()
    out: Dict[str, np.ndarray] = {}
    off = 0
    while off < len(blob):
        magic, version, typ, size, block = struct.unpack_from("<4siiii", blob, off)
        if magic != b"DNNw" or version != 0:
            raise ValueError(f"bad DNNw record header at byte {off}")
        name = blob[off + 20: off + _HDR].split(b"\0", 1)[0].decode("ascii")
        payload = blob[off + _HDR: off + _HDR + size]
        if typ == _TYPE_F32:
            arr = np.frombuffer(payload, dtype="<f4")
        elif typ == _TYPE_I32:
            arr = np.frombuffer(payload, dtype="<i4")
        elif typ == _TYPE_I8:
            arr = np.frombuffer(payload, dtype=np.int8)
        else:
            raise ValueError(f"record {name}: unknown type {typ}")
        out[name] = arr.copy()
        off += _HDR + block
    return out


def _dense_float(rec, name):
    """`*_weights_float` is W.T, i.e. (n_in, n_out) row-major."""
    bias = rec[name + "_bias"]
    n_out = bias.shape[0]
    w = rec[name + "_weights_float"].reshape(-1, n_out).T
    return np.ascontiguousarray(w, dtype=np.float32), bias.astype(np.float32)


def _dense_int8(rec, name):
    """8x4-blocked int8: stored as q.reshape(n_in/4,4,n_out/8,8).transpose(2,0,3,1)."""
    scale = rec[name + "_scale"].astype(np.float32) * np.float32(127.0)
    n_out = scale.shape[0]
    q = rec[name + "_weights_int8"]
    n_in = q.size // n_out
    q = q.reshape(n_out // 8, n_in // 4, 8, 4).transpose(1, 3, 0, 2).reshape(n_in, n_out)
    w = (q.astype(np.float32) * scale[None, :]).T
    return np.ascontiguousarray(w, dtype=np.float32), rec[name + "_bias"].astype(np.float32)


def _sparse_int8(rec, name, n_in):
    """GRU-input "sparse" form (common.py:140-176): per 8-output group an idx list [count, pos...]; each kept
    block is 4 inputs x 8 outputs stored output-major (8x4); blocks that are not listed are zero.  n_in is NOT in the
    blob (the reference compiles it in: `linear_init(..., nb_inputs, nb_outputs)`, common.py:274) -- a trailing input
    block that no group keeps leaves no trace in the index list -- so it comes from the architecture."""
    scale = rec[name + "_scale"].astype(np.float32) * np.float32(127.0)
    n_out = scale.shape[0]
    idx = rec[name + "_weights_idx"]
    q = rec[name + "_weights_int8"]
    wq = np.zeros((n_in, n_out), dtype=np.float32)
    p = 0
    blk = 0
    for g in range(n_out // 8):
        cnt = int(idx[p]); p += 1
        for _ in range(cnt):
            j = int(idx[p]); p += 1
            if j < 0 or j + 4 > n_in or j % 4:
                raise ValueError(f"{name}: block position {j} outside the {n_in} inputs of this layer")
            b = q[blk * 32:(blk + 1) * 32].reshape(8, 4).T  # (4 in, 8 out)
            wq[j:j + 4, g * 8:(g + 1) * 8] = b
            blk += 1
    if p != idx.size or blk * 32 != q.size:
        raise ValueError(f"{name}: index list and block data disagree")
    w = (wq * scale[None, :]).T
    return np.ascontiguousarray(w, dtype=np.float32), rec[name + "_bias"].astype(np.float32)


def _unswap_gates(a):
    """exporter order z,r,n -> torch order r,z,n (swap first two thirds)."""
    n = a.shape[0] // 3
    out = a.copy()
    out[0:n] = a[n:2 * n]
    out[n:2 * n] = a[0:n]
    return out


@dataclass
class GRU:
    w_ih: np.ndarray  # (3H, In) torch gate order r,z,n
    w_hh: np.ndarray  # (3H, H)
    b_ih: np.ndarray
    b_hh: np.ndarray


@dataclass
class Conv:
    w: np.ndarray  # (out, in, 2) torch layout; k=0 is the older tap
    b: np.ndarray
    dilation: int


@dataclass
class Dense:
    w: np.ndarray  # (out, in)
    b: np.ndarray


def _gru(rec, name, n_in):
    w_ih, b_ih = _sparse_int8(rec, name + "_input", n_in)
    w_hh, b_hh = _dense_int8(rec, name + "_recurrent")
    return GRU(_unswap_gates(w_ih), _unswap_gates(w_hh), _unswap_gates(b_ih), _unswap_gates(b_hh))


def _conv(rec, name, dilation):
    w, b = _dense_int8(rec, name)  # (out, 2*in), column index = k*in + i
    n_out, two_in = w.shape
    w3 = w.reshape(n_out, 2, two_in // 2).transpose(0, 2, 1)
    return Conv(np.ascontiguousarray(w3), b, dilation)


@dataclass
class Model:
    enc_dense1: Dense
    enc_gru: list
    enc_conv: list
    enc_zdense: Dense
    dec_dense1: Dense
    dec_gru: list
    dec_glu: list
    dec_conv: list
    dec_output: Dense


ENC_DILATION = (1, 2, 2, 2, 2)  # radae_base.py:241-249
# GRU input widths = the running DenseNet concat (radae_base.py:239-249, :377-391; rade_enc_data.h / rade_dec_data.h)
ENC_GRU_IN = tuple(64 + k * (64 + 96) for k in range(5))
DEC_GRU_IN = tuple(96 + k * (96 + 32) for k in range(5))


def load_model(path: str) -> Model:
    rec = read_records(path)
    return Model(
        enc_dense1=Dense(*_dense_float(rec, "enc_dense1")),
        enc_gru=[_gru(rec, f"enc_gru{i}", ENC_GRU_IN[i - 1]) for i in range(1, 6)],
        enc_conv=[_conv(rec, f"enc_conv{i}", ENC_DILATION[i - 1]) for i in range(1, 6)],
        enc_zdense=Dense(*_dense_float(rec, "enc_zdense")),
        dec_dense1=Dense(*_dense_float(rec, "dec_dense1")),
        dec_gru=[_gru(rec, f"dec_gru{i}", DEC_GRU_IN[i - 1]) for i in range(1, 6)],
        dec_glu=[Dense(*_dense_int8(rec, f"dec_glu{i}")) for i in range(1, 6)],
        dec_conv=[_conv(rec, f"dec_conv{i}", 1) for i in range(1, 6)],
        dec_output=Dense(*_dense_float(rec, "dec_output")),
    )


# ---------------------------------------------------------------------------------------------------
# Writer: fp32 Model -> DNNw blob, i.e. the forward direction of the reference's export
# (export_rade_weights.py:54-172 + wexchange/c_export/common.py + src/write_rade_weights.c:51-74).
# Used to give weight-less configurations (BBFM: no checkpoint exists) a deployable blob.
# ---------------------------------------------------------------------------------------------------
def _record(name: str, arr: np.ndarray) -> bytes:
    if arr.dtype == np.float32:
        typ = _TYPE_F32
    elif arr.dtype == np.int32:
        typ = _TYPE_I32
    elif arr.dtype == np.int8:
        typ = _TYPE_I8
    else:
        raise ValueError(arr.dtype)
    payload = arr.tobytes()
    block = (len(payload) + 63) // 64 * 64
    hdr = struct.pack("<4siiii", b"DNNw", 0, typ, len(payload), block) + name.encode("ascii").ljust(44, b"\0")[:44]
    return hdr + payload + b"\0" * (block - len(payload))


def _scaling(w_io: np.ndarray) -> np.ndarray:
    """common.py:180-194 on a float32 (n_in, n_out) matrix, in float32 like the exporter."""
    n_in = w_io.shape[0]
    m_abs = np.max(np.abs(w_io), axis=0)
    m_sum = np.max(np.abs(w_io[:n_in:2] + w_io[1:n_in:2]), axis=0)
    return np.maximum(m_abs / 127, m_sum / 129)


def _quant(w_io, scale):
    """common.py:132-137 (float32 division, round half to even, the exporter's bounds check)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        q = np.round(w_io / scale).astype("int")
    if q.max() > 127 or q.min() <= -128:
        raise ValueError("value out of bounds in quantize_weight (an all-zero output column, or a scale rule violated)")
    return q


def _emit_dense_float(out, name, w_oi, b):
    out.append(_record(name + "_weights_float", np.ascontiguousarray(w_oi.T, dtype=np.float32).ravel()))
    out.append(_record(name + "_bias", b.astype(np.float32)))


def _emit_int8(out, name, w_io, b, sparse):
    """One quantised linear layer as `print_linear_layer` (common.py:200-277) + the compiler + write_rade_weights.c make it.
    w_io: float32 (n_in, n_out) IN THE MEMORY ORDER THE EXPORTER HOLDS IT (a transposed view for dense / GRU matrices, a C-ordered copy for
    convolutions): `np.sum(q * scale, axis=0)` runs in float64 and its summation order follows the memory order, and byte identity of
    `subias` with the reference-exported blob (tests/golden/dnnw_export.npz) depends on it.  b: float32 or None."""
    assert w_io.dtype == np.float32
    n_in, n_out = w_io.shape
    assert n_in % 4 == 0 and n_out % 8 == 0
    scale = _scaling(w_io)                                            # float32
    q = _quant(w_io, scale)
    if sparse:   # common.py:140-176: every 4x8 block with a non-zero FLOAT weight kept, block stored output-major (8x4)
        idx, blocks = [], []
        for i in range(n_out // 8):
            pos = len(idx); idx.append(-1); cnt = 0
            for j in range(n_in // 4):
                blk = w_io[j * 4:(j + 1) * 4, i * 8:(i + 1) * 8]
                if np.sum(np.abs(blk)) > 1e-10:
                    cnt += 1; idx.append(j * 4)
                    blocks.append(q[j * 4:(j + 1) * 4, i * 8:(i + 1) * 8].T.reshape(-1))
            idx[pos] = cnt
        out.append(_record(name + "_weights_int8", np.concatenate(blocks).astype(np.int8)))
        out.append(_record(name + "_weights_idx", np.array(idx, dtype=np.int32)))
    else:        # common.py:59-69
        qq = q.reshape(n_in // 4, 4, n_out // 8, 8).transpose(2, 0, 3, 1)
        out.append(_record(name + "_weights_int8", np.ascontiguousarray(qq).astype(np.int8).ravel()))
    # common.py:263-268: int64 q x float32 scale -> float64 products, float64 sums; the C compiler then rounds the printed doubles to float
    subias = (np.zeros(n_out) if b is None else b.astype(np.float32)) - np.sum(q * scale, axis=0)
    out.append(_record(name + "_subias", subias.astype(np.float64).astype(np.float32)))
    out.append(_record(name + "_scale", (scale / 127).astype(np.float32)))
    out.append(_record(name + "_bias", (np.zeros(n_out) if b is None else b).astype(np.float32)))


def _swap_gates(a):
    n = a.shape[0] // 3
    out = a.copy(); out[0:n] = a[n:2 * n]; out[n:2 * n] = a[0:n]
    return out


def _emit_gru(out, name, g):
    """print_gru_layer (common.py:346-382): gates r,z,n -> z,r,n in place on the torch-oriented arrays, then the transposed VIEW."""
    _emit_int8(out, name + "_input", np.ascontiguousarray(_swap_gates(g.w_ih), dtype=np.float32).T, _swap_gates(g.b_ih), sparse=True)
    _emit_int8(out, name + "_recurrent", np.ascontiguousarray(_swap_gates(g.w_hh), dtype=np.float32).T, _swap_gates(g.b_hh), sparse=False)


def _emit_conv(out, name, c):
    """print_conv1d_layer (common.py:297-311): (out, in, k) -> (k, in, out) -> a C-ordered (k * in, out) copy."""
    w = np.transpose(np.asarray(c.w, dtype=np.float32), (2, 1, 0))
    _emit_int8(out, name, np.reshape(w, (-1, w.shape[-1])), c.b, sparse=False)


def write_blob(model: Model, path: str) -> None:
    """Quantise (int8 + per-output scale, like the reference's exporter) and write a DNNw blob."""
    out = []
    _emit_dense_float(out, "enc_dense1", model.enc_dense1.w, model.enc_dense1.b)
    _emit_dense_float(out, "enc_zdense", model.enc_zdense.w, model.enc_zdense.b)
    for i, g in enumerate(model.enc_gru, 1):
        _emit_gru(out, f"enc_gru{i}", g)
    for i, c in enumerate(model.enc_conv, 1):
        _emit_conv(out, f"enc_conv{i}", c)
    _emit_dense_float(out, "dec_dense1", model.dec_dense1.w, model.dec_dense1.b)
    for i, g in enumerate(model.dec_glu, 1):
        _emit_int8(out, f"dec_glu{i}", np.ascontiguousarray(g.w, dtype=np.float32).T, None, sparse=False)
    _emit_dense_float(out, "dec_output", model.dec_output.w, model.dec_output.b)
    for i, g in enumerate(model.dec_gru, 1):
        _emit_gru(out, f"dec_gru{i}", g)
    for i, c in enumerate(model.dec_conv, 1):
        _emit_conv(out, f"dec_conv{i}", c)
    with open(path, "wb") as f:
        f.write(b"".join(out))
