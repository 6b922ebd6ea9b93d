"""ctypes bindings for oracle/libradeoracle.so (TEST INFRASTRUCTURE ONLY -- see rade_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libradeoracle.so")
REPO = os.path.dirname(_HERE)
BLOB = os.path.join(REPO, "weights", "model19_check3.bin")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rade_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


class C32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class Trace(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("state", "nin", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "mf")] + \
               [(n, C.c_double) for n in ("fmax", "Dthresh", "Dtmax12", "Dtmax12_eoo")] + [("snrdB_3k_est", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orc_model_load.restype = vp; L.orc_model_load.argtypes = [C.c_char_p]
        L.orc_model_free.argtypes = [vp]
        L.orc_model_tensor.argtypes = [vp, C.c_char_p, C.POINTER(fp), ip]
        L.orc_get_const.argtypes = [C.c_char_p, vp, C.c_int]
        for n in ("orc_enc_new", "orc_dec_new", "orc_bpf_new"):
            getattr(L, n).restype = vp
        for n in ("orc_enc_reset", "orc_dec_reset", "orc_enc_free", "orc_dec_free", "orc_bpf_free", "orc_tx_free", "orc_rx_free"):
            getattr(L, n).argtypes = [vp]
        L.orc_core_encoder.argtypes = [vp, vp, vp, vp]
        L.orc_core_decoder.argtypes = [vp, vp, vp, vp]
        L.orc_core_encoder_b1.argtypes = [vp, vp, vp, vp]
        L.orc_model_feat_width.argtypes = [vp]
        L.orc_channel_rs.argtypes = [vp, vp, vp, vp, C.c_int, C.c_float]
        L.orc_channel_bbfm.argtypes = [vp, vp, vp, vp, C.c_int, C.c_float, C.c_float]
        L.orc_enc_gru_state.restype = fp; L.orc_enc_gru_state.argtypes = [vp, C.c_int]
        L.orc_dec_gru_state.restype = fp; L.orc_dec_gru_state.argtypes = [vp, C.c_int]
        L.orc_tx_new.restype = vp; L.orc_tx_new.argtypes = [vp]
        L.orc_tx_frame.argtypes = [vp, vp, vp, vp]
        L.orc_ofdm_mod.argtypes = [vp, vp]
        L.orc_tx_set_eoo_bits.argtypes = [vp, vp]
        L.orc_tx_eoo.argtypes = [vp, vp]
        L.orc_tx_set_txbpf.argtypes = [vp, C.c_int]
        L.orc_channel.argtypes = [vp, vp, C.c_int, vp, vp, C.c_float, C.c_float, C.c_float, vp]
        L.orc_channel_eoo.argtypes = [vp, vp, C.c_int, vp, C.c_float, C.c_float, C.c_float, C32]
        L.orc_sigma_from_EbNodB.restype = C.c_float; L.orc_sigma_from_EbNodB.argtypes = [C.c_float]
        L.orc_rx_new.restype = vp; L.orc_rx_new.argtypes = [vp]
        L.orc_rx_set_lcg.argtypes = [vp, C.c_uint]
        L.orc_rx_set_foff_err.argtypes = [vp, C.c_double]
        L.orc_rx_set_disable_unsync.argtypes = [vp, C.c_double]
        for n in ("orc_rx_nin", "orc_rx_sync", "orc_rx_snr"):
            getattr(L, n).argtypes = [vp]
        L.orc_rx_frame.argtypes = [vp, vp, vp, vp, vp]
        L.orc_rx_get_trace.argtypes = [vp, C.POINTER(Trace)]
        L.orc_bpf_run.argtypes = [vp, vp, vp, C.c_int]
        L.orc_distortion_loss.restype = C.c_double; L.orc_distortion_loss.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
        L.orc_find_loss.restype = C.c_double; L.orc_find_loss.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, ip]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


class Model:
    def __init__(self, path: str = BLOB):
        self.h = lib().orc_model_load(path.encode())
        if not self.h:
            raise OSError(f"oracle: cannot load {path}")

    def tensor(self, name):
        p = C.POINTER(C.c_float)(); n = C.c_int()
        if lib().orc_model_tensor(self.h, name.encode(), C.byref(p), C.byref(n)) != 0:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()


def get_const(name, n_floats, complex_=False):
    out = np.zeros(n_floats, np.float32)
    r = lib().orc_get_const(name.encode(), _p(out), n_floats)
    if r < 0:
        raise KeyError(name)
    out = out[:r]
    return out.view(np.complex64) if complex_ else out


class Encoder:
    def __init__(self, model):
        self.m = model; self.s = lib().orc_enc_new()

    def step(self, feat84, bottleneck=3):
        z = np.zeros(80, np.float32); f = f32(feat84)
        assert f.size == lib().orc_model_feat_width(self.m.h)
        (lib().orc_core_encoder_b1 if bottleneck == 1 else lib().orc_core_encoder)(self.m.h, self.s, _p(z), _p(f)); return z

    def gru_state(self, layer):
        return np.ctypeslib.as_array(lib().orc_enc_gru_state(self.s, layer), shape=(64,)).copy()


class Decoder:
    def __init__(self, model):
        self.m = model; self.s = lib().orc_dec_new()

    def step(self, z80):
        o = np.zeros(lib().orc_model_feat_width(self.m.h), np.float32); z = f32(z80)
        lib().orc_core_decoder(self.m.h, self.s, _p(o), _p(z)); return o

    def gru_state(self, layer):
        return np.ctypeslib.as_array(lib().orc_dec_gru_state(self.s, layer), shape=(96,)).copy()


class Tx:
    def __init__(self, model):
        self.m = model; self.h = lib().orc_tx_new(model.h)

    def frame(self, feat432):
        out = np.zeros(960, np.complex64); z = np.zeros(240, np.float32); f = f32(feat432)
        lib().orc_tx_frame(self.h, _p(out), _p(f), _p(z)); return out, z

    def set_eoo_bits(self, bits):
        b = f32(bits); lib().orc_tx_set_eoo_bits(self.h, _p(b))

    def set_txbpf(self, enable=True):
        lib().orc_tx_set_txbpf(self.h, int(enable))

    def eoo(self):
        out = np.zeros(1152, np.complex64); lib().orc_tx_eoo(self.h, _p(out)); return out


def channel_rs(z, H, noise, sigma):
    z = f32(z).ravel(); out = np.zeros_like(z)
    lib().orc_channel_rs(_p(out), _p(z), _p(f32(H).ravel()) if H is not None else None, _p(f32(noise).ravel()) if noise is not None else None, z.size, sigma)
    return out


def channel_bbfm(z, H, noise, CNRdB, Gfm):
    z = f32(z).ravel(); out = np.zeros_like(z)
    lib().orc_channel_bbfm(_p(out), _p(z), _p(f32(H).ravel()) if H is not None else None, _p(f32(noise).ravel()) if noise is not None else None, z.size, CNRdB, Gfm)
    return out


def ofdm_mod(z240):
    out = np.zeros(960, np.complex64); z = f32(z240); lib().orc_ofdm_mod(_p(out), _p(z)); return out


def channel(tx, G, noise, sigma, freq_offset, df_dt=0.0):
    tx = c64(tx); n = len(tx); rx = np.zeros(n, np.complex64); fin = np.zeros(1, np.complex64)
    G = c64(G) if G is not None else None; noise = c64(noise) if noise is not None else None
    lib().orc_channel(_p(rx), _p(tx), n, _p(G), _p(noise), sigma, freq_offset, df_dt, _p(fin))
    return rx, fin[0]


def channel_eoo(eoo, noise, sigma, freq_offset, df_dt, final_phase):
    eoo = c64(eoo); n = len(eoo); rx = np.zeros(n, np.complex64); noise = c64(noise) if noise is not None else None
    fp = C32(float(np.real(final_phase)), float(np.imag(final_phase)))
    lib().orc_channel_eoo(_p(rx), _p(eoo), n, _p(noise), sigma, freq_offset, df_dt, fp)
    return rx


class Bpf:
    def __init__(self):
        self.h = lib().orc_bpf_new()

    def run(self, x):
        x = c64(x); out = np.zeros(len(x), np.complex64); lib().orc_bpf_run(self.h, _p(out), _p(x), len(x)); return out


class Rx:
    def __init__(self, model, lcg_seed=1, foff_err=0.0, disable_unsync=0.0):
        self.m = model; self.h = lib().orc_rx_new(model.h)
        lib().orc_rx_set_lcg(self.h, lcg_seed); lib().orc_rx_set_foff_err(self.h, foff_err); lib().orc_rx_set_disable_unsync(self.h, disable_unsync)

    def nin(self):
        return lib().orc_rx_nin(self.h)

    def sync(self):
        return lib().orc_rx_sync(self.h)

    def snr(self):
        return lib().orc_rx_snr(self.h)

    def frame(self, rx_in):
        x = c64(rx_in); feat = np.zeros(432, np.float32); eoo = np.zeros(180, np.float32); z = np.zeros(240, np.float32)
        ret = lib().orc_rx_frame(self.h, _p(feat), _p(eoo), _p(x), _p(z))
        return ret, feat, eoo, z

    def trace(self):
        t = Trace(); lib().orc_rx_get_trace(self.h, C.byref(t)); return t


def run_rx_stream(model, stream, lcg_seed=1, foff_err=0.0, disable_unsync=0.0):
    """Drive the oracle receiver like radae_rxe.py:349-356; returns the same trace dict layout as
    oracle/gen_golden.py:run_rx."""
    rx = Rx(model, lcg_seed, foff_err, disable_unsync)
    keys_i = ["state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]
    keys_f = ["fmax", "Dthresh", "Dtmax12", "Dtmax12_eoo", "snrdB_3k_est"]
    tr = {k: [] for k in keys_i + keys_f}
    z_hat, feats, eoos = [], [], []
    pos = 0
    stream = c64(stream)
    while pos + rx.nin() <= len(stream):
        nin = rx.nin(); sb = rx.trace().state
        ret, f, e, z = rx.frame(stream[pos:pos + nin]); pos += nin
        t = rx.trace()
        for k, v in (("state_before", sb), ("state_after", t.state), ("nin_before", nin), ("nin_after", t.nin), ("ret", ret), ("tmax", t.tmax),
                     ("f_ind_max", t.f_ind_max), ("valid_count", t.valid_count), ("uw_errors", t.uw_errors), ("synced_count", t.synced_count),
                     ("snr_int", rx.snr()), ("fmax", t.fmax), ("Dthresh", t.Dthresh), ("Dtmax12", t.Dtmax12), ("Dtmax12_eoo", t.Dtmax12_eoo),
                     ("snrdB_3k_est", t.snrdB_3k_est)):
            tr[k].append(v)
        if ret & 1:
            z_hat.append(z); feats.append(f)
        if ret & 2:
            eoos.append(e)
    d = {k: np.array(tr[k], np.int32) for k in keys_i}
    d.update({k: np.array(tr[k], np.float64) for k in keys_f})
    d["z_hat"] = np.array(z_hat, np.float32).reshape(-1, 240)
    d["features_out"] = np.array(feats, np.float32).reshape(-1, 432)
    d["eoo_out"] = np.array(eoos, np.float32).reshape(-1, 180)
    return d


def distortion_loss(a, b, dim=20):
    a = f32(a); b = f32(b); stride = a.shape[-1]
    return lib().orc_distortion_loss(_p(a), _p(b), a.shape[0], dim, stride)


def find_loss(features, features_hat):
    a = f32(features); b = f32(features_hat); st = C.c_int()
    l = lib().orc_find_loss(_p(a), a.shape[0], _p(b), b.shape[0], a.shape[1], C.byref(st))
    return l, st.value
