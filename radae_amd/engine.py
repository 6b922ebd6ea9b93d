"""Python binding of the batched HIP engine (include/rade_batch.h) over ctypes.

PyTorch is used only as plumbing: device tensors own the HBM buffers whose raw pointers are handed
to the C ABI, and `torch.cuda.current_stream()` supplies the hipStream_t.  All compute happens in
radae_amd/libradehip.so (radae_amd/csrc); there is no CPU or PyTorch fallback -- if the shared
library or a GPU is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RADE_LIBRADEHIP") or os.path.join(_HERE, "libradehip.so")      # the override is a developer aid (A/B builds, tools/ab_build.sh)
DEFAULT_BLOB = os.path.join(os.path.dirname(_HERE), "weights", "model19_check3.bin")

NMF, NEOO, NIN_MAX, FEAT_MF, NEOO_BITS, ZMF = 960, 1152, 1120, 432, 180, 240
BOTTLENECK1, TX_BPF, BYPASS_DEC = 0x100, 0x400, 0x800          # include/rade_batch.h flags


class BatchConfig(C.Structure):
    _fields_ = [("n_streams", C.c_int), ("max_tx_mf", C.c_int), ("device", C.c_int), ("flags", C.c_int), ("rx_trace_calls", C.c_int), ("disable_unsync", C.c_float)]


class ChannelParams(C.Structure):
    _fields_ = [("n_sig", C.c_int), ("n_pre", C.c_int), ("n_post", C.c_int), ("with_eoo", C.c_int), ("sigma", C.c_float), ("freq_offset", C.c_float),
                ("df_dt", C.c_float), ("G_dev", C.c_void_p), ("noise_dev", C.c_void_p), ("seed", C.c_ulonglong),
                ("sine_amp", C.c_float), ("sine_freq", C.c_float), ("rx_gain", C.c_float)]


class RxStatus(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("consumed", "n_calls", "n_valid", "has_eoo", "nin", "sync", "snr_dB", "state")]


class RxTrace(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count",
                                        "uw_errors", "synced_count", "snr_int", "pad")] + \
               [(n, C.c_double) for n in ("fmax", "Dthresh", "Dtmax12", "Dtmax12_eoo")] + [("snrdB_3k_est", C.c_float), ("pad2", C.c_float)]


_lib = None


def load_library() -> C.CDLL:
    """dlopen radae_amd/libradehip.so and declare the C ABI.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"or `make -C radae_amd/csrc` (there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rade_batch_open.restype = vp; L.rade_batch_open.argtypes = [C.c_char_p, C.POINTER(BatchConfig)]
    L.rade_batch_open_mem.restype = vp; L.rade_batch_open_mem.argtypes = [vp, C.c_size_t, C.POINTER(BatchConfig)]
    L.rade_batch_close.argtypes = [vp]
    L.rade_batch_n_streams.argtypes = [vp]
    L.rade_batch_tx.argtypes = [vp, vp, C.c_int, vp, C.c_long, vp, vp]
    L.rade_batch_tx_latents.argtypes = [vp, vp, C.c_int, vp, C.c_long, vp]
    L.rade_batch_tx_channel.argtypes = [vp, vp, C.c_int, vp, C.c_long, vp, C.c_long, vp, vp]
    L.rade_batch_tx_set_eoo_bits.argtypes = [vp, vp]
    L.rade_batch_tx_eoo.argtypes = [vp, vp, C.c_long, vp]
    L.rade_batch_tx_reset.argtypes = [vp]
    L.rade_batch_channel.argtypes = [vp, vp, C.c_long, vp, C.c_long, C.POINTER(ChannelParams), vp]
    L.rade_batch_multipath_gen.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, vp, C.c_ulonglong, vp, vp]
    L.rade_batch_multipath_h.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, vp, vp]
    L.rade_sigma_from_EbNodB.restype = C.c_float; L.rade_sigma_from_EbNodB.argtypes = [C.c_float]
    L.rade_batch_rx.argtypes = [vp, vp, C.c_long, C.POINTER(C.c_int), C.c_int, vp, C.c_long, vp, C.POINTER(RxStatus), vp]
    L.rade_batch_rx_reset.argtypes = [vp]
    L.rade_batch_encode.argtypes = [vp, vp, C.c_int, vp, vp]
    L.rade_batch_decode.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
    L.rade_batch_channel_symbol.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_ulonglong, vp]
    L.rade_batch_reset.argtypes = [vp, vp]
    L.rade_batch_profile.argtypes = [vp, C.c_int]
    L.rade_batch_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.rade_batch_rx_set_lcg.argtypes = [vp, C.POINTER(C.c_uint)]
    L.rade_batch_rx_get_trace.argtypes = [vp, C.c_int, C.POINTER(RxTrace), vp, C.c_int]
    L.rade_batch_rx_stream_cycles.argtypes = [vp, vp]
    if hasattr(L, "rade_sync_policy"):
        L.rade_host_cpu_quota.restype = C.c_double; L.rade_host_cpu_quota.argtypes = []
        L.rade_sync_policy.argtypes = [C.c_int, C.c_double]
        L.rade_batch_sync_counts.argtypes = [vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]; L.rade_batch_sync_counts.restype = None
    if hasattr(L, "rade_batch_rx_filtered"):      # (absent from older A/B builds loaded through $RADE_LIBRADEHIP)
        L.rade_batch_rx_filtered.argtypes = [vp, C.c_int, vp, C.c_int]
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    # include/rade_api.h
    "rade_initialize", "rade_finalize", "rade_open", "rade_close", "rade_version", "rade_n_tx_out", "rade_n_tx_eoo_out", "rade_nin_max",
    "rade_n_features_in_out", "rade_n_eoo_bits", "rade_tx", "rade_tx_set_eoo_bits", "rade_tx_eoo", "rade_nin", "rade_rx", "rade_sync",
    "rade_freq_offset", "rade_snrdB_3k_est",
    # include/rade_batch.h
    "rade_batch_open", "rade_batch_open_mem", "rade_batch_close", "rade_batch_n_streams", "rade_batch_tx", "rade_batch_tx_latents", "rade_batch_tx_set_eoo_bits",
    "rade_batch_tx_eoo", "rade_batch_tx_reset", "rade_batch_channel", "rade_batch_tx_channel", "rade_batch_multipath_gen", "rade_batch_multipath_h", "rade_sigma_from_EbNodB", "rade_batch_rx", "rade_batch_rx_reset",
    "rade_batch_rx_set_lcg", "rade_batch_rx_get_trace", "rade_batch_reset", "rade_batch_profile", "rade_batch_profile_get", "rade_batch_profile_ref", "rade_batch_profile_intervals",
    "rade_batch_encode", "rade_batch_decode", "rade_batch_channel_symbol",
    "rade_batch_rx_stream_cycles", "rade_batch_rx_filtered", "rade_host_cpu_quota", "rade_sync_policy", "rade_batch_sync_counts",
    "rade_multi_open", "rade_multi_close", "rade_multi_n_devices", "rade_multi_transport", "rade_multi_engine", "rade_multi_shard", "rade_multi_foreach",
    "rade_multi_allreduce_sum",
]


def sigma_from_EbNodB(EbNodB: float) -> float:
    return float(load_library().rade_sigma_from_EbNodB(EbNodB))


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class BatchEngine:
    """B independent RADE streams on one GPU (one engine per process/GPU)."""

    def __init__(self, n_streams: int, max_tx_mf: int = 1, device: int = 0, flags: int = 0, blob: Optional[str] = None,
                 blob_bytes: Optional[bytes] = None, rx_trace_calls: int = 0, disable_unsync: float = 0.0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("radae_amd.BatchEngine needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.lib = load_library()
        self.B = n_streams
        self.device = torch.device("cuda", device)
        self.trace_calls = rx_trace_calls
        self.rx_row_floats = ZMF if flags & BYPASS_DEC else FEAT_MF      # RADE_BATCH_BYPASS_DEC: 240 latents per valid modem frame instead of 432 feature floats
        cfg = BatchConfig(n_streams, max_tx_mf, device, flags, rx_trace_calls, disable_unsync)
        if blob_bytes is not None:
            buf = C.create_string_buffer(blob_bytes, len(blob_bytes))
            self.h = self.lib.rade_batch_open_mem(C.cast(buf, C.c_void_p), len(blob_bytes), C.byref(cfg))
        else:
            self.h = self.lib.rade_batch_open((blob or DEFAULT_BLOB).encode(), C.byref(cfg))
        if not self.h:
            raise RuntimeError("rade_batch_open failed (see stderr)")
        self.max_tx_mf = max_tx_mf

    def close(self):
        if getattr(self, "h", None):
            self.lib.rade_batch_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        """Stream-ordered reset of encoder and receiver state (a new batch of utterances)."""
        self.lib.rade_batch_reset(self.h, _stream_ptr())

    PROF_CLASSES = ("gemm", "gru_scan", "ofdm_mod", "channel", "rx_sync", "rx_bpf")

    def profile(self, enable: bool):
        self.lib.rade_batch_profile(self.h, int(enable))

    def profile_ref(self, event_handle: int):
        """absolute launch intervals (profile_intervals) are measured from this hipEvent_t (e.g. torch.cuda.Event(enable_timing=True).cuda_event after record())"""
        self.lib.rade_batch_profile_ref.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.rade_batch_profile_ref(self.h, C.c_void_p(event_handle))

    def profile_intervals(self, cls_name: str, max_n: int = 4096):
        self.lib.rade_batch_profile_intervals.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        t0 = np.zeros(max_n, np.float32); t1 = np.zeros(max_n, np.float32)
        n = self.lib.rade_batch_profile_intervals(self.h, self.PROF_CLASSES.index(cls_name), t0.ctypes.data_as(C.c_void_p), t1.ctypes.data_as(C.c_void_p), max_n)
        return t0[:n].astype(np.float64), t1[:n].astype(np.float64)

    def profile_get(self):
        out = {}
        for i, name in enumerate(self.PROF_CLASSES):
            ms, wk, n = C.c_double(), C.c_double(), C.c_long()
            self.lib.rade_batch_profile_get(self.h, i, C.byref(ms), C.byref(wk), C.byref(n))
            out[name] = {"ms": ms.value, "flops": wk.value, "launches": n.value}
        return out

    # ---- transmit ---------------------------------------------------------------------------
    def tx(self, features, want_z: bool = False):
        """features: cuda float32 [B, n_mf*12, 36] -> iq complex64 [B, n_mf*960] (+ z [B, n_mf*3, 80])."""
        import torch
        assert features.is_cuda and features.dtype == torch.float32 and features.is_contiguous()
        B, nfr, w = features.shape
        assert B == self.B and w == 36 and nfr % 12 == 0
        n_mf = nfr // 12
        iq = torch.empty((B, n_mf * NMF), dtype=torch.complex64, device=features.device)
        z = torch.empty((B, n_mf * 3, 80), dtype=torch.float32, device=features.device) if want_z else None
        done = 0
        while done < n_mf:                       # chunk by the engine's capacity; state carries across chunks
            k = min(self.max_tx_mf, n_mf - done)
            f = features[:, done * 12:(done + k) * 12, :].contiguous()
            zc = torch.empty((B, k * 3, 80), dtype=torch.float32, device=features.device) if want_z else None
            r = self.lib.rade_batch_tx(self.h, f.data_ptr(), k, iq.data_ptr() + done * NMF * 8, n_mf * NMF, zc.data_ptr() if want_z else None, _stream_ptr())
            if r != k * NMF:
                raise RuntimeError("rade_batch_tx failed")
            if want_z:
                z[:, done * 3:(done + k) * 3] = zc
            done += k
        return (iq, z) if want_z else iq

    def tx_latents(self, z):
        """`radae_txe.py --bypass_enc` (radae_txe.py:124-126): z cuda float32 [B, n_mf*3, 80] from an external core encoder -> iq complex64 [B, n_mf*960]."""
        import torch
        assert z.is_cuda and z.dtype == torch.float32 and z.is_contiguous() and z.shape[0] == self.B and z.shape[2] == 80 and z.shape[1] % 3 == 0
        n_mf = z.shape[1] // 3
        iq = torch.empty((self.B, n_mf * NMF), dtype=torch.complex64, device=z.device)
        done = 0
        while done < n_mf:
            k = min(self.max_tx_mf, n_mf - done)
            zc = z[:, done * 3:(done + k) * 3, :].contiguous()
            if self.lib.rade_batch_tx_latents(self.h, zc.data_ptr(), k, iq.data_ptr() + done * NMF * 8, n_mf * NMF, _stream_ptr()) != k * NMF:
                raise RuntimeError("rade_batch_tx_latents failed")
            done += k
        return iq

    def tx_reset(self):
        self.lib.rade_batch_tx_reset(self.h)

    def set_eoo_bits(self, bits: Optional[np.ndarray]):
        if bits is None:
            r = self.lib.rade_batch_tx_set_eoo_bits(self.h, None)
        else:
            b = np.ascontiguousarray(bits, dtype=np.float32).reshape(self.B, NEOO_BITS)
            r = self.lib.rade_batch_tx_set_eoo_bits(self.h, b.ctypes.data_as(C.c_void_p))
        if r:
            raise RuntimeError("rade_batch_tx_set_eoo_bits failed")

    def tx_eoo(self):
        import torch
        out = torch.empty((self.B, NEOO), dtype=torch.complex64, device=self.device)
        if self.lib.rade_batch_tx_eoo(self.h, out.data_ptr(), NEOO, _stream_ptr()) != NEOO:
            raise RuntimeError("rade_batch_tx_eoo failed")
        return out

    # ---- core encoder / decoder alone, symbol-domain channels (configs 1, 2, 5) -----------------
    def encode(self, features):
        """features cuda float32 [B, n_steps, 4*feat_dim] -> z [B, n_steps, 80] (state carried; tx_reset() clears)."""
        import torch
        assert features.is_cuda and features.dtype == torch.float32 and features.is_contiguous() and features.shape[0] == self.B
        n = features.shape[1]
        z = torch.empty((self.B, n, 80), dtype=torch.float32, device=features.device)
        if self.lib.rade_batch_encode(self.h, features.data_ptr(), n, z.data_ptr(), _stream_ptr()) != n:
            raise RuntimeError("rade_batch_encode failed (n_steps > 3*max_tx_mf?)")
        return z

    def decode(self, z, feat_width: int, reset: bool = True):
        """z cuda float32 [B, n_steps, 80] -> features [B, n_steps, feat_width] (feat_width = 4*feat_dim of the blob)."""
        import torch
        assert z.is_cuda and z.dtype == torch.float32 and z.is_contiguous() and tuple(z.shape[::2]) == (self.B, 80)
        n = z.shape[1]
        out = torch.empty((self.B, n, feat_width), dtype=torch.float32, device=z.device)
        if self.lib.rade_batch_decode(self.h, z.data_ptr(), n, out.data_ptr(), int(reset), _stream_ptr()) != n:
            raise RuntimeError("rade_batch_decode failed")
        return out

    def channel_symbol(self, z, mode: str, p0: float, p1: float = 0.0, H=None, noise=None, seed: int = 0):
        """mode 'rs': z*H + sigma*noise (p0 = sigma, H per QPSK symbol [B, n*40]); mode 'bbfm': FM-demod SNR model
        (p0 = CNRdB, p1 = Gfm dB, H per real symbol [B, n*80]).  noise float32 [B, n*80] or None -> Philox(seed)."""
        import torch
        n = z.shape[1]
        out = torch.empty_like(z)
        r = self.lib.rade_batch_channel_symbol(self.h, z.data_ptr(), H.data_ptr() if H is not None else None, noise.data_ptr() if noise is not None else None,
                                               out.data_ptr(), n, 0 if mode == "rs" else 1, p0, p1, seed, _stream_ptr())
        if r != n:
            raise RuntimeError("rade_batch_channel_symbol failed")
        return out

    # ---- channel ----------------------------------------------------------------------------
    def channel(self, tx, sigma: float, freq_offset: float = 0.0, n_pre: int = 0, n_post: int = 0, with_eoo: bool = False,
                G=None, noise=None, seed: int = 0, df_dt: float = 0.0, sine_amp: float = 0.0, sine_freq: float = 0.0, rx_gain: float = 1.0):
        """tx complex64 [B, n_sig]; G complex64 [B, n_sig, 2] or None; noise complex64 [B, n_total] or None;
        sine_amp/sine_freq: complex tone over the whole output, rx_gain: final scale (inference.py:285-289)."""
        import torch
        assert tx.is_cuda and tx.dtype == torch.complex64 and tx.is_contiguous() and tx.shape[0] == self.B
        n_sig = tx.shape[1]
        n_total = n_pre + n_sig + (NEOO if with_eoo else 0) + n_post
        rx = torch.empty((self.B, n_total), dtype=torch.complex64, device=tx.device)
        p = ChannelParams(n_sig, n_pre, n_post, int(with_eoo), sigma, freq_offset, df_dt, None, None, seed, sine_amp, sine_freq, rx_gain)
        if G is not None:
            assert G.is_cuda and G.dtype == torch.complex64 and G.is_contiguous() and tuple(G.shape) == (self.B, n_sig, 2)
            p.G_dev = G.data_ptr()
        if noise is not None:
            assert noise.is_cuda and noise.dtype == torch.complex64 and noise.is_contiguous() and tuple(noise.shape) == (self.B, n_total)
            p.noise_dev = noise.data_ptr()
        r = self.lib.rade_batch_channel(self.h, tx.data_ptr(), n_sig, rx.data_ptr(), n_total, C.byref(p), _stream_ptr())
        if r != n_total:
            raise RuntimeError("rade_batch_channel failed")
        return rx

    def tx_channel(self, features, sigma: float, freq_offset: float = 0.0, n_pre: int = 0, n_post: int = 0, with_eoo: bool = False,
                   G=None, noise=None, seed: int = 0, df_dt: float = 0.0, want_iq: bool = False):
        """Transmit and channel in one pass (RADAE.forward): features [B, n_mf*12, 36] -> rx complex64 [B, n_total] (and iq if wanted).
        With G the modulator applies the two-path model itself (rade_batch_tx_channel); the whole utterance must fit max_tx_mf."""
        import torch
        assert features.is_cuda and features.dtype == torch.float32 and features.is_contiguous()
        B, nfr, w = features.shape
        assert B == self.B and w == 36 and nfr % 12 == 0 and nfr // 12 <= self.max_tx_mf
        n_mf = nfr // 12; n_sig = n_mf * NMF
        n_total = n_pre + n_sig + (NEOO if with_eoo else 0) + n_post
        rx = torch.empty((B, n_total), dtype=torch.complex64, device=features.device)
        iq = torch.empty((B, n_sig), dtype=torch.complex64, device=features.device) if (want_iq or G is None) else None
        p = ChannelParams(n_sig, n_pre, n_post, int(with_eoo), sigma, freq_offset, df_dt, None, None, seed, 0.0, 0.0, 1.0)
        if G is not None:
            assert G.is_cuda and G.dtype == torch.complex64 and G.is_contiguous() and tuple(G.shape) == (B, n_sig, 2)
            p.G_dev = G.data_ptr()
        if noise is not None:
            assert noise.is_cuda and noise.dtype == torch.complex64 and noise.is_contiguous() and tuple(noise.shape) == (B, n_total)
            p.noise_dev = noise.data_ptr()
        r = self.lib.rade_batch_tx_channel(self.h, features.data_ptr(), n_mf, iq.data_ptr() if iq is not None else None, n_sig, rx.data_ptr(), n_total, C.byref(p), _stream_ptr())
        if r != n_total:
            raise RuntimeError("rade_batch_tx_channel failed")
        return (rx, iq) if want_iq else rx

    def multipath_gen(self, channel: str, n_out: int, seed: int = 1, noise_low=None, fs: int = 8000):
        """Doppler-spread samples G [B, n_out, 2] complex64 generated on the device (multipath_samples.m presets
        mpg / mpp / mpd).  noise_low: optional complex64 [B, 2, n_low + 100] unit-variance-per-component low-rate
        input noise (parity with channel_tools.multipath_g); otherwise Philox from `seed`."""
        import torch
        from .channel_tools import PRESETS, doppler_plan
        taps, ratio, n_low = doppler_plan(PRESETS[channel][0], fs, n_out)
        G = torch.empty((self.B, n_out, 2), dtype=torch.complex64, device=self.device)
        tp = np.ascontiguousarray(taps, dtype=np.float32)
        nz = None
        if noise_low is not None:
            assert noise_low.is_cuda and noise_low.dtype == torch.complex64 and noise_low.is_contiguous() and tuple(noise_low.shape) == (self.B, 2, n_low + len(tp))
            nz = noise_low.data_ptr()
        r = self.lib.rade_batch_multipath_gen(self.h, tp.ctypes.data_as(C.POINTER(C.c_float)), len(tp), ratio, n_out, nz, seed, G.data_ptr(), _stream_ptr())
        if r != n_out:
            raise RuntimeError("rade_batch_multipath_gen failed")
        return G

    def multipath_h_gen(self, channel: str, n_sym: int, rs: int = 2000, nc: int = 1, fs: int = 8000, seed: int = 1, noise_low=None, complex_: bool = False):
        """multipath_samples.m's H output generated on the device: |H| float32 [B, n_sym, nc] (complex64 with complex_) at symbol rate rs from the preset's
        Doppler process at fs (BBFM.md:37: multipath_h_gen("lmr60", 20000) is `multipath_samples("lmr60", 8000, 2000, 1, 10, ...)` per stream)."""
        import torch
        from .channel_tools import PRESETS
        m = fs // rs
        assert m * rs == fs
        n_g = (n_sym - 1) * m + 1
        G = self.multipath_gen(channel, n_g, seed=seed, noise_low=noise_low, fs=fs)
        H = torch.empty((self.B, n_sym, nc, 2) if complex_ else (self.B, n_sym, nc), dtype=torch.float32, device=self.device)
        if self.lib.rade_batch_multipath_h(self.h, G.data_ptr(), n_g, m, n_sym, nc, PRESETS[channel][1], float(rs), int(complex_), H.data_ptr(), _stream_ptr()) != n_sym:
            raise RuntimeError("rade_batch_multipath_h failed")
        return torch.view_as_complex(H) if complex_ else H

    # ---- receive ----------------------------------------------------------------------------
    def rx_reset(self, lcg_seeds: Optional[Sequence[int]] = None):
        if lcg_seeds is None:
            self.lib.rade_batch_rx_reset(self.h)
        else:
            arr = (C.c_uint * self.B)(*[int(s) for s in lcg_seeds])
            self.lib.rade_batch_rx_set_lcg(self.h, arr)

    def rx(self, rx, n_avail=None, max_calls: int = 1 << 20, features_out=None, eoo_out=None):
        """rx complex64 [B, N] holding each stream's not-yet-consumed samples.  Returns
        (features [B, cap, 432] -- [B, cap, 240] latents for an engine opened with BYPASS_DEC --, status list[RxStatus], eoo [B, 180]).  features_out / eoo_out: caller-owned device buffers (as a C host passes
        them: rows beyond status.n_valid, and the EOO bits of a stream without status.has_eoo, keep whatever they held); without them fresh zeroed
        ones are allocated per call."""
        import torch
        assert rx.is_cuda and rx.dtype == torch.complex64 and rx.is_contiguous() and rx.shape[0] == self.B
        N = rx.shape[1]
        avail = np.full(self.B, N, np.int32) if n_avail is None else np.ascontiguousarray(n_avail, dtype=np.int32)
        cap = min(max_calls, int(avail.max()) // 800 + 1)
        if features_out is None:
            features_out = torch.zeros((self.B, cap, self.rx_row_floats), dtype=torch.float32, device=rx.device)
        # features_out.shape[1] is the per-stream frame capacity handed to the C ABI: a stream pauses (status.consumed < avail)
        # once it has filled its rows, so a short buffer never makes the kernel write into the next stream's region
        assert features_out.is_cuda and features_out.dtype == torch.float32 and features_out.is_contiguous() and features_out.dim() == 3 \
            and features_out.shape[0] == self.B and features_out.shape[1] >= 1 and features_out.shape[2] == self.rx_row_floats
        eoo = eoo_out if eoo_out is not None else torch.zeros((self.B, NEOO_BITS), dtype=torch.float32, device=rx.device)
        assert eoo.is_cuda and eoo.dtype == torch.float32 and eoo.is_contiguous() and tuple(eoo.shape) == (self.B, NEOO_BITS)
        status = (RxStatus * self.B)()
        r = self.lib.rade_batch_rx(self.h, rx.data_ptr(), N, avail.ctypes.data_as(C.POINTER(C.c_int)), max_calls, features_out.data_ptr(),
                                   features_out.shape[1] * self.rx_row_floats, eoo.data_ptr(), status, _stream_ptr())
        if r:
            raise RuntimeError("rade_batch_rx failed")
        return features_out, list(status), eoo

    def rx_filtered(self, b: int, n: int) -> np.ndarray:
        """The first n band-pass filtered samples stream b's receiver read in the most recent rx() invocation (complex_bpf.bpf output)."""
        out = np.zeros(n, np.complex64)
        got = self.lib.rade_batch_rx_filtered(self.h, b, out.ctypes.data_as(C.c_void_p), n)
        if got < 0:
            raise RuntimeError("rade_batch_rx_filtered failed")
        return out[:got]

    def sync_counts(self):
        """(waits that slept on the blocking event, waits that spun) of this engine's rx() calls so far"""
        a, b = C.c_long(0), C.c_long(0)
        if not hasattr(self.lib, "rade_batch_sync_counts"):      # (older A/B builds loaded through $RADE_LIBRADEHIP)
            return 0, 0
        self.lib.rade_batch_sync_counts(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def rx_stream_cycles(self) -> np.ndarray:
        """Shader-clock cycles each stream's workgroup spent in the most recent receiver launch."""
        out = np.zeros(self.B, np.int64)
        if self.lib.rade_batch_rx_stream_cycles(self.h, out.ctypes.data_as(C.c_void_p)) != self.B:
            raise RuntimeError("rade_batch_rx_stream_cycles failed")
        return out

    def rx_trace(self, b: int = 0):
        """Per-call trace of stream b in the layout of tests/golden/rxtrace_*.npz."""
        n = self.trace_calls
        tr = (RxTrace * n)()
        z = np.zeros((n, ZMF), np.float32)
        got = self.lib.rade_batch_rx_get_trace(self.h, b, tr, z.ctypes.data_as(C.c_void_p), n)
        if got < 0:
            raise RuntimeError("trace not enabled")
        ints = ["state_before", "state_after", "nin_before", "nin_after", "ret", "tmax", "f_ind_max", "valid_count", "uw_errors", "synced_count", "snr_int"]
        flts = ["fmax", "Dthresh", "Dtmax12", "Dtmax12_eoo", "snrdB_3k_est"]
        d = {k: np.array([getattr(tr[i], k) for i in range(got)], np.int32) for k in ints}
        d.update({k: np.array([getattr(tr[i], k) for i in range(got)], np.float64) for k in flts})
        d["z_all"] = z[:got]
        d["z_hat"] = z[:got][(d["ret"] & 1) == 1]
        d["eoo_out"] = z[:got][(d["ret"] & 2) == 2][:, :NEOO_BITS]
        return d
