"""Host-side mirror of the reference's streaming classes over the C ABI (include/rade_api.h).

`radae_tx` / `radae_rx` keep the method names, argument meaning and return conventions of
/root/reference/radae_txe.py:47-144 and /root/reference/radae_rxe.py:56-330 (the classes
rade_api.c drives through CPython), but every call goes through libradehip.so's `rade_*` C entry
points -- i.e. the same boundary freedv-gui / radae_tx.c / radae_rx.c bind to -- and runs on the GPU.
One `rade_open()` handle carries one Tx and one Rx, as in the reference ("single context only").
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import engine

RADE_USE_C_ENCODER, RADE_USE_C_DECODER, RADE_FOFF_TEST, RADE_VERBOSE_0 = 0x1, 0x2, 0x4, 0x8
RADE_BATCH_TX_BPF = 0x400          # include/rade_batch.h: honoured by rade_open() too (the reference's txbpf_en)


def _bind():
    L = engine.load_library()
    if getattr(L, "_rade_api_bound", False):
        return L
    vp = C.c_void_p
    L.rade_open.restype = vp; L.rade_open.argtypes = [C.c_char_p, C.c_int]
    L.rade_close.argtypes = [vp]
    for n in ("rade_n_tx_out", "rade_n_tx_eoo_out", "rade_nin_max", "rade_n_features_in_out", "rade_n_eoo_bits", "rade_nin", "rade_sync", "rade_snrdB_3k_est"):
        getattr(L, n).argtypes = [vp]
    L.rade_freq_offset.restype = C.c_float; L.rade_freq_offset.argtypes = [vp]
    L.rade_tx.argtypes = [vp, vp, vp]
    L.rade_tx_set_eoo_bits.argtypes = [vp, vp]
    L.rade_tx_eoo.argtypes = [vp, vp]
    L.rade_rx.argtypes = [vp, vp, C.POINTER(C.c_int), vp, vp]
    L._rade_api_bound = True
    return L


class Rade:
    """Thin RAII wrapper of `struct rade *`."""

    def __init__(self, model_file: str = "", flags: int = RADE_VERBOSE_0):
        self.L = _bind()
        self.L.rade_initialize()
        self.flags = flags
        self.r = self.L.rade_open(model_file.encode(), flags)
        if not self.r:
            raise RuntimeError("rade_open failed: no GPU or no weight blob (libradehip.so has no CPU fallback)")

    def close(self):
        if getattr(self, "r", None):
            self.L.rade_close(self.r); self.r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class radae_tx:
    """radae_txe.radae_tx: 12 feature frames x 36 floats in -> 960 IQ samples out per call.  `txbpf_en` as in radae_txe.py:47-83: every frame and
    the end-of-over frame through the Tx band-pass filter and the magnitude clip (a handle passed in must have been opened with RADE_BATCH_TX_BPF)."""

    def __init__(self, model_name: str = "", handle: Rade | None = None, flags: int = RADE_VERBOSE_0, txbpf_en: bool = False):
        if txbpf_en:
            flags |= RADE_BATCH_TX_BPF
        if handle is not None and bool(handle.flags & RADE_BATCH_TX_BPF) != bool(txbpf_en):
            # the reference's radae_tx always honours txbpf_en; here the filter belongs to the handle, so a mismatch would silently transmit the other signal
            raise ValueError(f"radae_tx(txbpf_en={bool(txbpf_en)}) on a handle opened {'with' if handle.flags & RADE_BATCH_TX_BPF else 'without'} RADE_BATCH_TX_BPF")
        self.h = handle or Rade(model_name, flags)
        self.txbpf_en = bool(txbpf_en)
        L, r = self.h.L, self.h.r
        self.n_floats_in = L.rade_n_features_in_out(r)
        self.Nmf = L.rade_n_tx_out(r)
        self.Neoo = L.rade_n_tx_eoo_out(r)

    def get_n_features_in(self): return self.n_floats_in
    def get_n_floats_in(self): return self.n_floats_in
    def get_Nmf(self): return self.Nmf
    def get_Neoo(self): return self.Neoo
    def get_Neoo_bits(self): return self.h.L.rade_n_eoo_bits(self.h.r)

    def set_eoo_bits(self, eoo_bits):
        b = np.ascontiguousarray(eoo_bits, dtype=np.float32)
        assert b.size == self.get_Neoo_bits()
        self.h.L.rade_tx_set_eoo_bits(self.h.r, b.ctypes.data_as(C.c_void_p))

    def do_radae_tx(self, buffer_f32, tx_out):
        f = np.ascontiguousarray(buffer_f32, dtype=np.float32)
        assert f.size == self.n_floats_in and tx_out.dtype == np.complex64 and tx_out.size == self.Nmf
        n = self.h.L.rade_tx(self.h.r, tx_out.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p))
        assert n == self.Nmf

    def do_eoo(self, tx_out):
        assert tx_out.dtype == np.complex64 and tx_out.size == self.Neoo
        n = self.h.L.rade_tx_eoo(self.h.r, tx_out.ctypes.data_as(C.c_void_p))
        assert n == self.Neoo


class radae_rx:
    """radae_rxe.radae_rx: `get_nin()` samples in per call; returns valid | endofover<<1."""

    def __init__(self, model_name: str = "", handle: Rade | None = None, flags: int = RADE_VERBOSE_0, foff_err: float = 0.0):
        if foff_err:
            flags |= RADE_FOFF_TEST      # the C ABI only offers the 10 Hz developer test (rade_api.c:263-264)
        self.h = handle or Rade(model_name, flags)
        L, r = self.h.L, self.h.r
        self.n_floats_out = L.rade_n_features_in_out(r)
        self._eoo = np.zeros(L.rade_n_eoo_bits(r), np.float32)

    def get_n_features_out(self): return self.n_floats_out
    def get_n_floats_out(self): return self.n_floats_out
    def get_nin_max(self): return self.h.L.rade_nin_max(self.h.r)
    def get_nin(self): return self.h.L.rade_nin(self.h.r)
    def get_sync(self): return bool(self.h.L.rade_sync(self.h.r))
    def get_snrdB_3k_est(self): return self.h.L.rade_snrdB_3k_est(self.h.r)
    def get_Neoo_bits(self): return self.h.L.rade_n_eoo_bits(self.h.r)

    def do_radae_rx(self, buffer_complex, floats_out):
        x = np.ascontiguousarray(buffer_complex, dtype=np.complex64)
        assert x.size >= self.get_nin() and floats_out.dtype == np.float32 and floats_out.size == self.n_floats_out
        has_eoo = C.c_int(0)
        n = self.h.L.rade_rx(self.h.r, floats_out.ctypes.data_as(C.c_void_p), C.byref(has_eoo), self._eoo.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
        if has_eoo.value:
            floats_out[:] = 0
            floats_out[:self._eoo.size] = self._eoo        # radae_rxe.py:321-323
        return (1 if n else 0) | (2 if has_eoo.value else 0)


# ---- the reference's bypass modes (radae_txe.py --bypass_enc, radae_rxe.py --bypass_dec): an external core encoder / decoder sits on the other side --------------------
# rade_api.h has no switch for them (the reference's rade_api.c uses them internally with its C core, :421-431, :491-513), so these two classes drive a one-stream
# batched engine (include/rade_batch.h: rade_batch_tx_latents, RADE_BATCH_BYPASS_DEC) behind the reference classes' method names.
class radae_tx_bypass_enc:
    """radae_txe.radae_tx(..., bypass_enc=True): 3 x 80 latents in (n_floats_in = 240, radae_txe.py:67-69) -> 960 IQ samples out per call."""

    def __init__(self, model_name: str = "", txbpf_en: bool = False, device: int = 0):
        import torch
        self.eng = engine.BatchEngine(1, max_tx_mf=2, device=device, blob=model_name or None, flags=engine.TX_BPF if txbpf_en else 0)
        self._torch, self.dev = torch, torch.device("cuda", device)
        self.n_floats_in, self.Nmf, self.Neoo, self.txbpf_en, self.bypass_enc = engine.ZMF, engine.NMF, engine.NEOO, bool(txbpf_en), True

    def get_n_floats_in(self): return self.n_floats_in
    def get_Nmf(self): return self.Nmf
    def get_Neoo(self): return self.Neoo
    def get_Neoo_bits(self): return engine.NEOO_BITS
    def set_eoo_bits(self, eoo_bits): self.eng.set_eoo_bits(np.ascontiguousarray(eoo_bits, dtype=np.float32).reshape(1, -1))

    def do_radae_tx(self, buffer_f32, tx_out):
        z = np.ascontiguousarray(buffer_f32, dtype=np.float32)
        assert z.size == self.n_floats_in and tx_out.dtype == np.complex64 and tx_out.size == self.Nmf
        tx_out[:] = self.eng.tx_latents(self._torch.tensor(z.reshape(1, 3, 80), device=self.dev)).cpu().numpy()[0]

    def do_eoo(self, tx_out):
        assert tx_out.dtype == np.complex64 and tx_out.size == self.Neoo
        tx_out[:] = self.eng.tx_eoo().cpu().numpy()[0]


class radae_rx_engine:
    """radae_rxe.radae_rx over a one-stream batched engine instead of the rade_api.h handle: the form that also offers what the C ABI has no switch for --
    `disable_unsync` (radae_rxe.py --disable_unsync) and, in the subclass below, `bypass_dec`.  Same methods and return convention as `radae_rx`."""
    _flags, _row = 0, engine.FEAT_MF

    def __init__(self, model_name: str = "", foff_err: float = 0.0, disable_unsync: float = 0.0, device: int = 0):
        import torch
        self.eng = engine.BatchEngine(1, max_tx_mf=1, device=device, blob=model_name or None, flags=self._flags | (RADE_FOFF_TEST if foff_err else 0), disable_unsync=disable_unsync)
        self._torch, self.dev = torch, torch.device("cuda", device)
        self.n_floats_out, self.bypass_dec = self._row, bool(self._flags & engine.BYPASS_DEC)
        self._nin, self._sync, self._snr = engine.NMF, 0, 0
        self._rows = torch.zeros((1, 1, self._row), dtype=torch.float32, device=self.dev)
        self._eoo = torch.zeros((1, engine.NEOO_BITS), dtype=torch.float32, device=self.dev)

    def get_n_features_out(self): return self.n_floats_out
    def get_n_floats_out(self): return self.n_floats_out
    def get_nin_max(self): return engine.NIN_MAX
    def get_nin(self): return self._nin
    def get_sync(self): return bool(self._sync)
    def get_snrdB_3k_est(self): return self._snr
    def get_Neoo_bits(self): return engine.NEOO_BITS

    def do_radae_rx(self, buffer_complex, floats_out):
        x = np.ascontiguousarray(buffer_complex, dtype=np.complex64)
        assert x.size >= self._nin and floats_out.dtype == np.float32 and floats_out.size == self.n_floats_out
        _, st, _ = self.eng.rx(self._torch.tensor(x[None, :self._nin], device=self.dev), max_calls=1, features_out=self._rows, eoo_out=self._eoo)
        s = st[0]
        assert s.n_calls == 1
        self._nin, self._sync, self._snr = s.nin, s.sync, s.snr_dB
        if s.n_valid:
            floats_out[:] = self._rows.cpu().numpy()[0, 0]
        if s.has_eoo:
            floats_out[:] = 0
            floats_out[:engine.NEOO_BITS] = self._eoo.cpu().numpy()[0]      # radae_rxe.py:321-323
        return (1 if s.n_valid else 0) | (2 if s.has_eoo else 0)


class radae_rx_bypass_dec(radae_rx_engine):
    """radae_rxe.radae_rx(..., bypass_dec=True): `get_nin()` samples in per call; a valid call writes the modem frame's 240 equalised latents (n_floats_out = 240,
    radae_rxe.py:121-123, :315), an end-of-over call the EOO soft bits; the UW errors are never summed (:300-312); returns valid | endofover << 1."""
    _flags, _row = engine.BYPASS_DEC, engine.ZMF
