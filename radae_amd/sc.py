"""Host-side mirror of the reference's `single_carrier` modem (radae/dsp.py:579-860) over the batched C ABI
(include/rade_batch.h, `rade_sc_*`; kernels in radae_amd/csrc/rade_sc.hip).  No CPU fallback: every call runs on the GPU
through libradehip.so."""
import ctypes as C

import numpy as np
import torch

from .engine import load_library


class ScStatus(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("consumed", C.c_int), ("state", C.c_int), ("nin", C.c_int), ("fs_s", C.c_int), ("g", C.c_float),
                ("max_cs_re", C.c_float), ("max_cs_im", C.c_float), ("norm_rx_timing", C.c_float), ("phase_ambiguity", C.c_float)]


class ScFrame(C.Structure):
    _fields_ = [("state", C.c_int), ("nin", C.c_int), ("fs_s", C.c_int), ("pad", C.c_int), ("norm_rx_timing", C.c_float), ("g", C.c_float),
                ("max_cs_re", C.c_float), ("max_cs_im", C.c_float), ("phase_ambiguity", C.c_float), ("pad2", C.c_float * 3)]


SC_SYMBOLS = ["rade_sc_open", "rade_sc_close", "rade_sc_reset", "rade_sc_n_streams", "rade_sc_n_tx_out", "rade_sc_nin_max", "rade_sc_n_payload",
              "rade_sc_rrc", "rade_sc_tx", "rade_sc_rx"]
_FRAME_DT = np.dtype([("state", "i4"), ("nin", "i4"), ("fs_s", "i4"), ("pad", "i4"), ("norm_rx_timing", "f4"), ("g", "f4"), ("max_cs_re", "f4"),
                      ("max_cs_im", "f4"), ("phase_ambiguity", "f4"), ("pad2", "f4", 3)])


def _lib():
    L = load_library()
    vp = C.c_void_p
    L.rade_sc_open.restype = vp; L.rade_sc_open.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
    L.rade_sc_close.argtypes = [vp]; L.rade_sc_reset.argtypes = [vp]
    for n in ("rade_sc_n_streams", "rade_sc_n_tx_out", "rade_sc_nin_max", "rade_sc_n_payload"):
        getattr(L, n).argtypes = [vp]
    L.rade_sc_rrc.argtypes = [vp, vp]
    L.rade_sc_tx.argtypes = [vp, vp, C.c_int, vp, C.c_long, vp]
    L.rade_sc_rx.argtypes = [vp, vp, C.c_long, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    return L


class SingleCarrierBatch:
    """B independent modems; arguments as single_carrier.__init__ (dsp.py:581)."""

    def __init__(self, n_streams, Rs=2400.0, Fs=9600.0, fcentreHz=0.0, alpha=0.25, device=0):
        self.L = _lib()
        self.B, self.dev = n_streams, torch.device("cuda", device)
        self.h = self.L.rade_sc_open(n_streams, Rs, Fs, fcentreHz, alpha, device)
        if not self.h:
            raise RuntimeError("rade_sc_open failed (no GPU? libradehip.so has no CPU fallback)")
        self.n_tx_out, self.nin_max, self.n_payload = self.L.rade_sc_n_tx_out(self.h), self.L.rade_sc_nin_max(self.h), self.L.rade_sc_n_payload(self.h)

    def close(self):
        if self.h:
            self.L.rade_sc_close(self.h); self.h = None

    def reset(self):
        self.L.rade_sc_reset(self.h)

    def rrc(self):
        t = np.zeros(24, np.float64); self.L.rade_sc_rrc(self.h, t.ctypes.data_as(C.c_void_p)); return t

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def tx(self, symbs: torch.Tensor) -> torch.Tensor:
        """symbs [B, n_frames, 80] float32 (cuda) -> [B, n_frames * 384] complex64 (dsp.py:636-662)."""
        assert symbs.is_cuda and symbs.dtype == torch.float32 and symbs.shape[0] == self.B and symbs.shape[2] == self.n_payload
        symbs = symbs.contiguous(); nfr = symbs.shape[1]
        out = torch.empty((self.B, nfr * self.n_tx_out), dtype=torch.complex64, device=self.dev)
        rc = self.L.rade_sc_tx(self.h, symbs.data_ptr(), nfr, out.data_ptr(), out.shape[1], self._stream())
        if rc < 0:
            raise RuntimeError("rade_sc_tx failed")
        return out

    def rx(self, rx: torch.Tensor, max_frames=None):
        """rx [B, n] complex64 (cuda): every whole frame available is demodulated (dsp.py:773-829).
        -> payload [B, F, 80] complex64, z_hat [B, F, 80] float32, frames (numpy structured [B, F]), status list."""
        assert rx.is_cuda and rx.dtype == torch.complex64 and rx.shape[0] == self.B
        rx = rx.contiguous(); n = rx.shape[1]
        F = max_frames or max(1, n // (self.n_tx_out - 1))
        pay = torch.zeros((self.B, F, self.n_payload), dtype=torch.complex64, device=self.dev)
        zh = torch.zeros((self.B, F, self.n_payload), dtype=torch.float32, device=self.dev)
        fr = torch.zeros((self.B, F, _FRAME_DT.itemsize), dtype=torch.uint8, device=self.dev)
        st = (ScStatus * self.B)()
        rc = self.L.rade_sc_rx(self.h, rx.data_ptr(), n, n, F, pay.data_ptr(), zh.data_ptr(), fr.data_ptr(), C.cast(st, C.c_void_p), self._stream())
        if rc != 0:
            raise RuntimeError("rade_sc_rx failed")
        frames = fr.cpu().numpy().view(_FRAME_DT).reshape(self.B, F)
        return pay, zh, frames, list(st)


# ---- the reference's stdin/stdout filters (sc_tx.py:58-75, sc_rx.py:83-112) over whole streams ----------------------
def sc_tx_stream(z: np.ndarray, fcentreHz=1500.0, scale=16384.0, real=True) -> np.ndarray:
    """[frames, 80] float32 latents -> int16 samples (real part, or I/Q interleaved with real=False)."""
    z = np.ascontiguousarray(z, np.float32).reshape(1, -1, 80)
    m = SingleCarrierBatch(1, fcentreHz=fcentreHz)
    tx = (scale * m.tx(torch.tensor(z, device=m.dev))).cpu().numpy()[0]
    m.close()
    if real:
        return tx.real.astype(np.int16)
    return np.stack([tx.real, tx.imag], axis=1).astype(np.int16).reshape(-1)


def sc_rx_stream(samples: np.ndarray, fcentreHz=1500.0, real=True) -> np.ndarray:
    """int16 samples -> [synced frames, 80] float32 z_hat = g Re(payload), one row per frame that ends in sync."""
    x = np.asarray(samples, np.int16)
    rx = x.astype(np.float32).astype(np.complex64) if real else (x[0::2].astype(np.float32) + 1j * x[1::2].astype(np.float32)).astype(np.complex64)
    m = SingleCarrierBatch(1, fcentreHz=fcentreHz)
    _, zh, fr, st = m.rx(torch.tensor(rx[None], device=m.dev))
    m.close()
    k = st[0].n_frames
    return zh.cpu().numpy()[0, :k][fr[0, :k]["state"] == 1]


def main(argv=None):
    import argparse, sys
    ap = argparse.ArgumentParser(description="single-carrier modem filters on the GPU (the reference's sc_tx.py / sc_rx.py)")
    ap.add_argument("mode", choices=["tx", "rx"])
    ap.add_argument("--fcentreHz", type=float, default=1500.0)
    ap.add_argument("--scale", type=float, default=16384.0)
    ap.add_argument("--complex", dest="real", action="store_false")
    a = ap.parse_args(argv)
    data = sys.stdin.buffer.read()
    if a.mode == "tx":
        n = len(data) // 320
        sys.stdout.buffer.write(sc_tx_stream(np.frombuffer(data[:n * 320], np.float32).reshape(n, 80), a.fcentreHz, a.scale, a.real).tobytes())
    else:
        sys.stdout.buffer.write(sc_rx_stream(np.frombuffer(data[:len(data) // 2 * 2], np.int16), a.fcentreHz, a.real).astype(np.float32).tobytes())


if __name__ == "__main__":
    main()
