"""CPU restatement of the reference's single-carrier BPSK modem for BBFM symbols -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/radae/dsp.py:532-562 (gen_rn_coeffs) and :579-860 (class single_carrier: tx, rx_Fs_to_Rs,
est_timing_and_decimate, est_phase_and_correct, rx) with the arithmetic types NumPy 2 gives that code (the
reference mixes complex64 buffers with float64 scalars, so most of the receiver runs in complex128).  Pinned
against tests/golden/sc_*.npz, which oracle/gen_golden_sc.py produced by importing the reference here.
Only tests/ may import this module; the product path is radae_amd/csrc/rade_sc.hip.
"""
import numpy as np

SYNC_WORD = np.array([1, 1, 1, 1, 1, -1, 1, 1, -1, -1, 1, 1, -1, -1, -1, -1], dtype=np.complex64)   # first 16 of the P25 word, dsp.py:592-593
NSYNC, NFRAME, NPAYLOAD, NFILT_SYM, NPHASE, SAMPLE_POINT = 16, 96, 80, 6, 21, 5


def rrc_coeffs(alpha, Rs, Fs, Nsym=NFILT_SYM):
    """dsp.py:532-562: root-Nyquist taps by spectral square root of a raised cosine (4096-point FFT)."""
    M = int(Fs / Rs); T = 1.0 / Fs; Ts = 1.0 / Rs
    n = np.arange(-Nsym * Ts / 2, Nsym * Ts / 2, T)
    den = np.pi * n / Ts
    sinc = np.where(np.abs(den) < 1e-10, 1.0, np.sin(np.pi * n / Ts) / np.where(np.abs(den) < 1e-10, 1.0, den))
    cden = 1 - (2 * alpha * n / Ts) ** 2
    cosop = np.where(np.abs(cden) < 1e-10, np.pi / 4, np.cos(alpha * np.pi * n / Ts) / np.where(np.abs(cden) < 1e-10, 1.0, cden))
    G = np.fft.fft(sinc * cosop, 4096) / M
    G = np.where(np.abs(G) < 0.02, G * 0.001, G)
    root = np.sqrt(np.abs(G)) * np.exp(1j * np.angle(G))
    return np.fft.ifft(root)[:Nsym * M].real


class SingleCarrier:
    """State and per-frame processing of dsp.py:581-633 (constructor), :636-662 (tx), :747-860 (rx)."""

    def __init__(self, Rs=2400, Fs=9600, fcentreHz=0.0, alpha=0.25):
        self.M = int(Fs / Rs); assert self.M == Fs / Rs
        self.Fs = Fs
        self.omega = 2 * np.pi * fcentreHz / Fs
        self.rrc = rrc_coeffs(alpha, Rs, Fs)
        self.Ntap = len(self.rrc)
        self.tx_mem = np.zeros(self.Ntap, np.complex64)
        self.rx_mem = np.zeros(self.Ntap, np.complex64)
        self.rx_filt_out = np.zeros((NFRAME + 2) * self.M, np.complex64)
        self.nin = NFRAME * self.M
        self.rx_symb_buf = np.zeros(2 * NFRAME, np.complex64)
        self.phase_mem = np.zeros(NPHASE, np.complex128)
        self.phase_fine = 0.0; self.phase_coarse = 0.0; self.phase_ambiguity = 0.0
        self.tx_lo = 1 + 0j; self.rx_lo = 1 + 0j
        self.state = "search"; self.fs_s = 0; self.g = 1.0; self.bad_fs = 0
        self.max_Cs = 0j; self.norm_rx_timing = 0.0

    def tx(self, symbs):
        """80 payload symbols -> 96*M samples (dsp.py:636-662)."""
        M, Nt = self.M, self.Ntap
        s = np.concatenate([SYNC_WORD, np.asarray(symbs).astype(np.complex64)])
        fin = np.concatenate([self.tx_mem, np.zeros(len(s) * M, np.complex64)])
        fin[Nt::M] = s * M
        n = len(s) * M
        idx = np.arange(n)[:, None] + 1 + np.arange(Nt)[None, :]
        out = (fin[idx].astype(np.complex128) @ self.rrc).astype(np.complex64)
        self.tx_mem = fin[-Nt:]
        lo = self.tx_lo * np.exp(1j * self.omega * np.arange(n))       # the reference multiplies the phasor up sample by sample
        out = (out * lo).astype(np.complex64)
        self.tx_lo = self.tx_lo * np.exp(1j * self.omega * n)
        self.tx_lo /= abs(self.tx_lo)
        return out

    def rx(self, x):
        """nin samples -> 80 phase-resolved payload symbols of the frame-sync position (dsp.py:747-860)."""
        M, Nt = self.M, self.Ntap
        assert len(x) == self.nin
        nin = self.nin
        bb = (np.asarray(x, np.complex64) * (self.rx_lo * np.exp(-1j * self.omega * np.arange(nin)))).astype(np.complex64)   # :753-758
        self.rx_lo = self.rx_lo * np.exp(-1j * self.omega * nin); self.rx_lo /= abs(self.rx_lo)
        fin = np.concatenate([self.rx_mem, bb])                                                                              # :761-766
        keep = len(self.rx_filt_out) - nin
        self.rx_filt_out[:keep] = self.rx_filt_out[-keep:].copy()
        idx = np.arange(nin)[:, None] + 1 + np.arange(Nt)[None, :]
        self.rx_filt_out[keep:] = (fin[idx].astype(np.complex128) @ self.rrc).astype(np.complex64)
        self.rx_mem = fin[-Nt:]
        # fine timing (:668-704): phase of the symbol-rate line of the envelope, linear interpolation at the best instant
        rf = self.rx_filt_out
        env = np.abs(rf[SAMPLE_POINT:]).astype(np.float64)
        xx = np.dot(env, np.exp(-1j * 2 * np.pi * np.arange(len(env)) / M))
        norm = np.angle(xx) / (2 * np.pi)
        corr = -norm * M
        low = int(np.floor(corr)); fract = corr - low
        samp = SAMPLE_POINT + low + np.arange(0, NFRAME * M, M)
        sym = rf[samp].astype(np.complex128) * (1 - fract) + rf[samp + 1].astype(np.complex128) * fract
        self.nin = NFRAME * M
        if norm < -0.35: self.nin += M / 4
        if norm > 0.35: self.nin -= M / 4
        self.nin = int(self.nin)
        self.norm_rx_timing = norm
        # phase (:707-742): BPSK stripped by squaring, 21-symbol window, pi jumps tracked
        buf = np.concatenate([self.phase_mem, sym])
        corrected = np.zeros(NFRAME, np.complex64)
        for s in range(NFRAME):
            acc = 0
            for v in buf[s + 1:s + 1 + NPHASE] ** 2: acc = acc + v
            fine = np.angle(acc) / 2
            if fine - self.phase_fine < -0.9 * np.pi: self.phase_coarse += np.pi
            if fine - self.phase_fine > 0.9 * np.pi: self.phase_coarse -= np.pi
            self.phase_fine = fine
            corrected[s] = buf[s + NPHASE // 2] * np.exp(-1j * (self.phase_coarse + fine))
        self.phase_mem = buf[-NPHASE:]
        self.rx_symb_buf[:NFRAME] = self.rx_symb_buf[NFRAME:].copy()
        self.rx_symb_buf[NFRAME:] = corrected
        # frame sync (:773-826)
        fs_s = self.fs_s; nxt = self.state; b = self.rx_symb_buf
        if self.state == "search":
            mx, ms = 0j, 0
            for s in range(NFRAME):
                r = b[s:s + NSYNC]
                num = np.dot(np.conj(r), SYNC_WORD / np.complex64(4))
                den = np.sqrt(np.dot(np.conj(r), r))
                Cs = num / (den + 1e-12)
                if abs(Cs) > abs(mx): mx, ms = Cs, s
            self.max_Cs = mx
            if abs(mx) >= 0.5:
                nxt = "sync"; fs_s = ms; self.fs_s = ms; self.bad_fs = 0
                self.phase_ambiguity = np.pi if mx.real < 0 else 0.0
                self.g = 1 / (np.mean(np.abs(b[fs_s:fs_s + NSYNC]) ** 2) ** 0.5 + 1e-12)
        if self.state == "sync":
            r = np.exp(1j * self.phase_ambiguity) * b[fs_s:fs_s + NSYNC]
            n_err = int(np.sum(r * SYNC_WORD < 0))
            self.bad_fs = self.bad_fs + 1 if n_err > 2 else 0
            if self.bad_fs >= 3: nxt = "search"
            self.g = 1 / (np.mean(np.abs(b[fs_s:fs_s + NSYNC]) ** 2) ** 0.5 + 1e-12)
        self.state = nxt
        return np.exp(1j * self.phase_ambiguity) * b[fs_s + NSYNC:fs_s + NFRAME]


def run_rx_stream(m, rx):
    """Feed a sample stream frame by frame (sc_rx.py:83-112); -> per-frame dict of arrays."""
    out = {k: [] for k in ("state", "nin", "fs_s", "norm_rx_timing", "g", "max_Cs", "phase_ambiguity", "payload")}
    n = 0
    while len(rx) - n >= m.nin:
        nin = m.nin
        pay = m.rx(rx[n:n + nin]); n += nin
        out["state"].append(1 if m.state == "sync" else 0); out["nin"].append(m.nin); out["fs_s"].append(m.fs_s)
        out["norm_rx_timing"].append(m.norm_rx_timing); out["g"].append(m.g); out["max_Cs"].append(complex(m.max_Cs))
        out["phase_ambiguity"].append(m.phase_ambiguity); out["payload"].append(np.asarray(pay, np.complex128))
    return {k: np.array(v) for k, v in out.items()}, n
