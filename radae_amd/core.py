"""ctypes mirror of include/rade_core.h: the rade_core_encoder / rade_core_decoder level of the C path
(/root/reference/src/rade_core.h:42-46; harnesses test_rade_enc.c / test_rade_dec.c), one 40 ms step per call,
host buffers in and out.  All compute happens in libradehip.so; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import DEFAULT_BLOB, load_library

CORE_SYMBOLS = ["rade_parse_weights", "init_radeenc", "init_radedec", "rade_init_encoder", "rade_core_encoder", "rade_init_decoder",
                "rade_core_decoder", "rade_free_encoder", "rade_reset_encoder", "rade_reset_decoder", "rade_free_decoder"]


class WeightArray(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int), ("size", C.c_int), ("data", C.c_void_p)]


class _Model(C.Structure):       # RADEEnc / RADEDec
    _fields_ = [("blob", C.c_void_p), ("blob_len", C.c_int), ("dim", C.c_int), ("nb_z", C.c_int)]


class _State(C.Structure):       # RADEEncState / RADEDecState
    _fields_ = [("initialized", C.c_int), ("dev", C.c_void_p)]


def _lib():
    L = load_library()
    L.rade_parse_weights.argtypes = [C.POINTER(C.POINTER(WeightArray)), C.c_void_p, C.c_int]
    L.init_radeenc.argtypes = [C.POINTER(_Model), C.POINTER(WeightArray), C.c_int]
    L.init_radedec.argtypes = [C.POINTER(_Model), C.POINTER(WeightArray), C.c_int]
    for n in ("rade_init_encoder", "rade_init_decoder", "rade_free_encoder", "rade_reset_encoder", "rade_reset_decoder", "rade_free_decoder"):
        getattr(L, n).argtypes = [C.POINTER(_State)]; getattr(L, n).restype = None
    L.rade_core_encoder.argtypes = [C.POINTER(_State), C.POINTER(_Model), C.c_void_p, C.c_void_p, C.c_int, C.c_int]; L.rade_core_encoder.restype = None
    L.rade_core_decoder.argtypes = [C.POINTER(_State), C.POINTER(_Model), C.c_void_p, C.c_void_p, C.c_int]; L.rade_core_decoder.restype = None
    return L


def parse_weights(blob: bytes):
    """[(name, type, size)] of a DNNw blob through the C walker (Opus parse_weights() role)."""
    L = _lib()
    buf = C.create_string_buffer(blob, len(blob))
    lst = C.POINTER(WeightArray)()
    n = L.rade_parse_weights(C.byref(lst), C.cast(buf, C.c_void_p), len(blob))
    if n < 0:
        raise ValueError("not a DNNw blob")
    return [(lst[i].name.decode(), lst[i].type, lst[i].size) for i in range(n)]


class _Core:
    def __init__(self, blob_path: str, dim: int, enc: bool):
        self.L = _lib()
        self._blob = open(blob_path or DEFAULT_BLOB, "rb").read()
        self._buf = C.create_string_buffer(self._blob, len(self._blob))       # stays mapped while the model is in use
        self._list = C.POINTER(WeightArray)()
        if self.L.rade_parse_weights(C.byref(self._list), C.cast(self._buf, C.c_void_p), len(self._blob)) < 0:
            raise ValueError("not a DNNw blob")
        self.model, self.state, self.dim, self.enc = _Model(), _State(), dim, enc
        if (self.L.init_radeenc if enc else self.L.init_radedec)(C.byref(self.model), self._list, dim) != 0:
            raise ValueError(f"blob does not hold a model with dimension {dim}")
        self.reset()

    def reset(self):
        if self.state.initialized:
            (self.L.rade_reset_encoder if self.enc else self.L.rade_reset_decoder)(C.byref(self.state))
        else:
            (self.L.rade_init_encoder if self.enc else self.L.rade_init_decoder)(C.byref(self.state))

    def close(self):
        (self.L.rade_free_encoder if self.enc else self.L.rade_free_decoder)(C.byref(self.state))


class CoreEncoder(_Core):
    def __init__(self, blob_path: str = DEFAULT_BLOB, input_dim: int = 84):
        super().__init__(blob_path, input_dim, True)

    def step(self, features, bottleneck: int = 3) -> np.ndarray:
        f = np.ascontiguousarray(features, dtype=np.float32).ravel()
        assert f.size == self.dim
        z = np.zeros(80, np.float32)
        self.L.rade_core_encoder(C.byref(self.state), C.byref(self.model), z.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p), 0, bottleneck)
        return z


class CoreDecoder(_Core):
    def __init__(self, blob_path: str = DEFAULT_BLOB, output_dim: int = 84):
        super().__init__(blob_path, output_dim, False)

    def step(self, z_hat) -> np.ndarray:
        z = np.ascontiguousarray(z_hat, dtype=np.float32).ravel()
        assert z.size == 80
        out = np.zeros(self.dim, np.float32)
        self.L.rade_core_decoder(C.byref(self.state), C.byref(self.model), out.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), 0)
        return out
