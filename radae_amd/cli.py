"""stdin / stdout filters with the command lines of the reference's Python tools:

    python -m radae_amd.cli txe [--txbpf] [--bypass_enc] [--eoo_data_test] [--model_name BLOB]     features.f32 (or z.f32) -> IQ.f32      /root/reference/radae_txe.py:146-180
    python -m radae_amd.cli rxe [--bypass_dec] [--disable_unsync S] [--foff_err HZ] [--eoo_data_test] [--no_stdout] [-v N] [--model_name BLOB]
                                                                                                     IQ.f32 -> features.f32 (or z_hat.f32)  /root/reference/radae_rxe.py:332-371

    python -m radae_amd.cli inference MODEL features.f32 features_hat.f32 --EbNodB .. [--g_file g.f32] --write_rx rx.f32 [...]    the channel-simulation run of inference.py (rate Fs)
    python -m radae_amd.cli multipath_samples mpp 8000 50 30 10 h.f32 g.f32                                                      multipath_samples.m
    python -m radae_amd.cli bbfm_inference MODEL features.f32 features_hat.f32 [--CNRdB ..] [--h_file h_lmr60.f32] [--write_latent z.f32]    bbfm_inference.py

so that the reference's shell pipelines (`cat features_in.f32 | python3 radae_txe.py > rx.f32`, `cat rx.f32 | python3 radae_rxe.py > features_out.f32`: CMakeLists.txt:300-420) run with
`python3 -m radae_amd.cli txe|rxe` in their place.  Everything computes in libradehip.so on the GPU (radae_amd/api.py over include/rade_api.h; the bypass modes over a one-stream
batched engine); `--model_name` takes a DNNw blob (the reference's `.pth` checkpoints are not in its tree), default weights/model19_check3.bin.  `--noauxdata` is not offered: model19_check3 has the aux symbol.
"""
from __future__ import annotations

import argparse
import struct
import sys

import numpy as np

from . import api


def _txe(argv):
    ap = argparse.ArgumentParser(prog="radae_amd.cli txe", description="RADAE streaming transmitter, features.f32 on stdin, IQ.f32 on output")
    ap.add_argument("--model_name", type=str, default="", help="DNNw weight blob (default: weights/model19_check3.bin)")
    ap.add_argument("--txbpf", action="store_true", help="enable Tx BPF")
    ap.add_argument("--bypass_enc", action="store_true", help="Bypass core encoder, read z from stdin")
    ap.add_argument("--eoo_data_test", action="store_true", help="experimental EOO data test - tx test frame")
    args = ap.parse_args(argv)
    tx = api.radae_tx_bypass_enc(args.model_name, txbpf_en=args.txbpf) if args.bypass_enc else api.radae_tx(args.model_name, txbpf_en=args.txbpf)
    if args.eoo_data_test:                                  # radae_txe.py:157-163: the seeded bits the receiver side regenerates
        rng = np.random.default_rng(65647)
        bits = np.sign(rng.random(tx.get_Neoo_bits()) - 0.5).astype(np.float32)
        tx.set_eoo_bits(bits)
        bits.tofile("eoo_tx.f32")
    n_in = tx.get_n_floats_in()
    tx_out = np.zeros(tx.get_Nmf(), np.complex64)
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        buf = inp.read(n_in * struct.calcsize("f"))
        if len(buf) != n_in * struct.calcsize("f"):
            break
        tx.do_radae_tx(np.frombuffer(buf, np.float32), tx_out)
        out.write(tx_out.tobytes())
    eoo = np.zeros(tx.get_Neoo(), np.complex64)
    tx.do_eoo(eoo)
    out.write(eoo.tobytes())
    out.flush()
    return 0


def _rxe(argv):
    ap = argparse.ArgumentParser(prog="radae_amd.cli rxe", description="RADAE streaming receiver, IQ.f32 on stdin to features.f32 on stdout")
    ap.add_argument("--model_name", type=str, default="", help="DNNw weight blob (default: weights/model19_check3.bin)")
    ap.add_argument("-v", type=int, default=2, help="Verbose level (default 2)")
    ap.add_argument("--disable_unsync", type=float, default=0.0, help="test mode: disable auxdata based unsyncs after this many seconds (default disabled)")
    ap.add_argument("--no_stdout", action="store_false", dest="use_stdout", help="disable the use of stdout")
    ap.add_argument("--foff_err", type=float, default=0.0, help="Artifical freq offset error after first sync to test false sync (the C ABI offers the 10 Hz test only)")
    ap.add_argument("--bypass_dec", action="store_true", help="Bypass core decoder, write z_hat to stdout")
    ap.add_argument("--eoo_data_test", action="store_true", help="experimental EOO data test - count bit errors")
    ap.set_defaults(use_stdout=True)
    args = ap.parse_args(argv)
    if args.bypass_dec or args.disable_unsync:              # both are switches of the batched engine (rade_api.h has neither)
        cls = api.radae_rx_bypass_dec if args.bypass_dec else api.radae_rx_engine
        rx = cls(args.model_name, foff_err=args.foff_err, disable_unsync=args.disable_unsync)
    else:
        rx = api.radae_rx(args.model_name, foff_err=args.foff_err)
    floats_out = np.zeros(rx.get_n_floats_out(), np.float32)
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    mf = 0
    while True:
        nin = rx.get_nin()
        buf = inp.read(nin * struct.calcsize("ff"))
        if len(buf) != nin * struct.calcsize("ff"):
            break
        ret = rx.do_radae_rx(np.frombuffer(buf, np.complex64), floats_out)
        mf += 1
        if args.v >= 2:
            print(f"{mf:3d} sync: {int(rx.get_sync())} nin: {rx.get_nin():4d} SNRdB: {rx.get_snrdB_3k_est():3d} ret: {ret}", file=sys.stderr)
        if (ret & 1) and args.use_stdout:
            out.write(floats_out.tobytes())
        if (ret & 2) and args.eoo_data_test:                # radae_rxe.py:359-368
            rng = np.random.default_rng(65647)
            tx_bits = np.sign(rng.random(rx.get_Neoo_bits()) - 0.5)
            n_bits = len(tx_bits)
            n_errors = int(np.sum(floats_out[:n_bits] * tx_bits < 0))
            ber = n_errors / n_bits
            print(f"EOO data n_bits: {n_bits} n_errors: {n_errors} BER: {ber:5.2f}", file=sys.stderr)
            if ber < 0.05:
                print("PASS", file=sys.stderr)
    out.flush()
    if args.v >= 1:
        print(f"state: {'sync' if rx.get_sync() else 'search'}", file=sys.stderr)       # what the ctest radae_rx_slip_plus_drops greps for
    return 0


def _inference(argv):
    """inference.py's rate-Fs channel-simulation run as the streaming ctests use it (`inference.sh model wav /dev/null --EbNodB .. --freq_offset .. [--df_dt ..] [--g_file g.f32]
    --rate_Fs --pilots --pilot_eq --eq_ls --cp 0.004 --bottleneck 3 --time_offset -16 --auxdata --write_rx rx.f32 [--prepend_noise s] [--append_noise s] [--end_of_over]
    [--sine_amp a --sine_freq f] [--rx_gain g] [--write_tx tx.f32]`, CMakeLists.txt:300-420; inference.py:43-79, :253-300): encoder + OFDM modulator, the two-path Doppler
    channel from a `g_file` (multipath_samples' format: gain sample, then ..G1G2..), AWGN at the Eb/No, frequency offset / drift, the write_rx tail.  The model-shape switches
    (--rate_Fs --pilots --pilot_eq --eq_ls --cp --bottleneck --time_offset --auxdata --correct_freq_offset --coarse_mag --latent-dim) are accepted and must describe model19_check3's
    waveform, the only one this path implements.  `features_hat` receives what the STREAMING receiver (radae_rxe's) decodes from the written samples -- the reference runs its
    stateless receiver with ideal timing there; the ctests of this path pass /dev/null.  Noise: the device's Philox generator (--seed), not torch's."""
    import torch
    from . import engine, wire
    from .loss import find_loss
    ap = argparse.ArgumentParser(prog="radae_amd.cli inference")
    ap.add_argument("model_name"); ap.add_argument("features"); ap.add_argument("features_hat")
    ap.add_argument("--EbNodB", type=float, default=100.0); ap.add_argument("--g_file", type=str, default=""); ap.add_argument("--write_rx", type=str, default="")
    ap.add_argument("--rx_gain", type=float, default=1.0); ap.add_argument("--write_tx", type=str, default=""); ap.add_argument("--freq_offset", type=float, default=0.0)
    ap.add_argument("--df_dt", type=float, default=0.0); ap.add_argument("--prepend_noise", type=float, default=0.0); ap.add_argument("--append_noise", type=float, default=0.0)
    ap.add_argument("--end_of_over", action="store_true"); ap.add_argument("--sine_amp", type=float, default=0.0); ap.add_argument("--sine_freq", type=float, default=1000.0)
    ap.add_argument("--loss_test", type=float, default=0.0); ap.add_argument("--seed", type=int, default=1)
    for flag in ("--rate_Fs", "--pilots", "--pilot_eq", "--eq_ls", "--auxdata", "--correct_freq_offset", "--coarse_mag"):
        ap.add_argument(flag, action="store_true")
    ap.add_argument("--cp", type=float, default=0.004); ap.add_argument("--bottleneck", type=int, default=3); ap.add_argument("--time_offset", type=int, default=-16)
    ap.add_argument("--latent-dim", type=int, default=80)
    args = ap.parse_args(argv)
    if args.bottleneck != 3 or abs(args.cp - 0.004) > 1e-9 or args.time_offset != -16 or args.latent_dim != 80:
        raise SystemExit("radae_amd.cli inference: only model19_check3's waveform (--rate_Fs --pilots --pilot_eq --eq_ls --cp 0.004 --bottleneck 3 --time_offset -16 --auxdata) is implemented")
    blob = args.model_name if args.model_name.endswith(".bin") else None
    feats = wire.read_features(args.features)
    n_mf = len(feats) // 12                                   # whole modem frames (radae.py:303-310)
    feats = np.ascontiguousarray(feats[:12 * n_mf])
    dev = torch.device("cuda", 0)
    eng = engine.BatchEngine(1, max_tx_mf=n_mf, blob=blob)
    iq = eng.tx(torch.tensor(feats[None], device=dev))
    n_sig = n_mf * engine.NMF
    G = None
    if args.g_file:
        g = np.fromfile(args.g_file, np.complex64).reshape(-1, 2)
        mp_gain = np.real(g[0, 0]); g = (mp_gain * g[1:]).astype(np.complex64)          # inference.py:160-171
        if len(g) < n_sig:
            raise SystemExit("Multipath Doppler spread file too short")
        G = torch.tensor(np.ascontiguousarray(g[:n_sig])[None], device=dev)
    sigma = engine.sigma_from_EbNodB(args.EbNodB)
    n_pre, n_post = int(8000 * args.prepend_noise), int(8000 * args.append_noise)
    rx = eng.channel(iq, sigma, args.freq_offset, n_pre=n_pre, n_post=n_post, with_eoo=args.end_of_over, G=G, seed=args.seed, df_dt=args.df_dt,
                     sine_amp=args.sine_amp, sine_freq=args.sine_freq, rx_gain=args.rx_gain)
    tx = iq.cpu().numpy()[0]
    S = float(np.mean(np.abs(tx) ** 2)); N = sigma ** 2
    EbNo = 10 ** (args.EbNodB / 10); Rb, Bw = 2000.0, 3000.0
    print("          Eb/No   C/No     SNR3k  Rb'    Eq     PAPR")
    print(f"Target..: {args.EbNodB:6.2f}  {10 * np.log10(EbNo * Rb):6.2f}  {10 * np.log10(EbNo * Rb / Bw):6.2f}  {2400:d}")
    cno = 10 * np.log10(S * 8000.0 / N)
    print(f"Measured: {cno + 10 * np.log10(160 / (8000.0 * 30 * 2)):6.2f}  {cno:6.2f}  {cno - 10 * np.log10(Bw):6.2f}                {20 * np.log10(np.max(np.abs(tx)) / np.sqrt(S)):5.2f}")
    if args.write_rx:
        rx.cpu().numpy()[0].astype(np.complex64).tofile(args.write_rx)
    if args.write_tx:
        tx.astype(np.complex64).tofile(args.write_tx)
    fo, st, _ = eng.rx(rx if args.rx_gain == 1.0 else rx.clone())
    nv = st[0].n_valid
    fh = fo.cpu().numpy()[0, :nv].reshape(-1, 36)
    if args.features_hat != "/dev/null":
        fh.astype(np.float32).tofile(args.features_hat)
    if nv:
        loss, start = find_loss(feats, fh)
        print(f"loss: {loss:5.3f} (streaming receiver: {12 * nv} frames decoded, aligned at frame {start})")
        if args.loss_test > 0.0:
            print("PASS" if loss < args.loss_test else "FAIL")
    else:
        print("loss: n/a (the streaming receiver decoded nothing)")
    eng.close()
    return 0


def _multipath_samples(argv):
    """multipath_samples.m's command line: `multipath_samples(ch, Fs, Rs, Nc, Nseconds, H_fn, G_fn="", H_complex=0)` -> the rate-Rs `H` file (magnitudes, or complex with
    --complex) and the rate-Fs `G` file (gain sample, then ..G1G2.. complex64) that inference.py's --h_file / --g_file read.  numpy's generator instead of Octave's randn('seed', 1)."""
    from .channel_tools import PRESETS
    ap = argparse.ArgumentParser(prog="radae_amd.cli multipath_samples")
    ap.add_argument("ch", choices=sorted(PRESETS)); ap.add_argument("Fs", type=int); ap.add_argument("Rs", type=int); ap.add_argument("Nc", type=int)
    ap.add_argument("Nseconds", type=float); ap.add_argument("H_fn"); ap.add_argument("G_fn", nargs="?", default="")
    ap.add_argument("--complex", action="store_true", dest="h_complex"); ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    import math
    from .channel_tools import doppler_spread
    nsam = int(args.Fs * args.Nseconds)
    rng = np.random.default_rng(args.seed)
    g1 = doppler_spread(PRESETS[args.ch][0], args.Fs, nsam, rng); g2 = doppler_spread(PRESETS[args.ch][0], args.Fs, nsam, rng)      # the draws of channel_tools.multipath_g
    hf_gain = 1.0 / math.sqrt(np.var(g1) + np.var(g2))                                 # multipath_samples.m:31
    m = args.Fs // args.Rs
    d = PRESETS[args.ch][1]
    H = hf_gain * (g1[::m, None] + g2[::m, None] * np.exp(-2j * np.pi * np.arange(args.Nc)[None, :] * d * args.Rs))      # :33-40
    (H.astype(np.complex64) if args.h_complex else np.abs(H).astype(np.float32)).tofile(args.H_fn)
    if args.G_fn:                                                                      # :92-103: four floats of hf_gain, then ..G1G2.. UN-scaled (inference.py:160-171 multiplies)
        out = np.concatenate([np.full((1, 2), hf_gain * (1 + 1j), np.complex64), np.stack([g1, g2], axis=1).astype(np.complex64)])
        out.tofile(args.G_fn)
    print(f"{args.ch}: Doppler spread {PRESETS[args.ch][0]:g} Hz, path delay {d * 1e3:g} ms, {len(H)} x {args.Nc} H samples" + (f", {nsam} G samples" if args.G_fn else ""))
    return 0


def _bbfm_inference(argv):
    """bbfm_inference.py (:43-170): features -> core encoder (bottleneck 1) -> the analog-FM channel model (bbfm.py:157-197: per-symbol CNR = 20 log10 |H| + CNRdB, FM demodulator SNR
    with its threshold at 12 dB, noise, clamp) -> core decoder -> features_hat; `--h_file` = rate-Rs fading magnitudes (multipath_samples("lmr60", 8000, 2000, 1, ...)), `--write_latent`,
    `--write_CNRdB`, `--loss_test`, `--passthru`.  `model_name`: a DNNw blob of the BBFM architecture (default weights/bbfm_random_seed20240501.bin: no trained BBFM weights exist in the
    reference tree).  Noise: the device's Philox generator (--seed)."""
    import math
    import os
    import torch
    from . import engine, wire
    from .loss import distortion_loss
    ap = argparse.ArgumentParser(prog="radae_amd.cli bbfm_inference")
    ap.add_argument("model_name"); ap.add_argument("features"); ap.add_argument("features_hat")
    ap.add_argument("--latent-dim", type=int, default=80); ap.add_argument("--write_latent", type=str, default=""); ap.add_argument("--CNRdB", type=float, default=100.0)
    ap.add_argument("--passthru", action="store_true"); ap.add_argument("--h_file", type=str, default=""); ap.add_argument("--write_CNRdB", type=str, default="")
    ap.add_argument("--loss_test", type=float, default=0.0); ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    feats = wire.read_features(args.features)
    T = len(feats) // 4                                      # encoder steps of four frames (bbfm.py / radae_base.py:158)
    feats = np.ascontiguousarray(feats[:4 * T])
    if args.passthru:
        feats.astype(np.float32).tofile(args.features_hat); return 0
    blob = args.model_name if args.model_name.endswith(".bin") else os.path.join(os.path.dirname(engine.DEFAULT_BLOB), "bbfm_random_seed20240501.bin")
    dev = torch.device("cuda", 0)
    Tc = 3 * ((T + 2) // 3)
    eng = engine.BatchEngine(1, max_tx_mf=Tc // 3, blob=blob, flags=engine.BOTTLENECK1)
    x = np.zeros((1, Tc, 80), np.float32); x[0, :T] = feats[:, :20].reshape(T, 80)
    z = eng.encode(torch.tensor(x, device=dev))
    nsym = Tc * 80
    H = None; Hn = np.ones(nsym, np.float32)
    if args.h_file:
        h = np.fromfile(args.h_file, np.float32)
        if h.size < T * 80:
            raise SystemExit("Multipath H file too short")
        Hn[:T * 80] = h[:T * 80]; H = torch.tensor(Hn[None], device=dev)
    Gfm = 10 * math.log10(3 * (5000 / 3000) ** 2 * (5000 / 3000 + 1))       # bbfm.py:78-80, fd 5000 Hz, fm 3000 Hz
    zh = eng.channel_symbol(z, "bbfm", args.CNRdB, Gfm, H=H, seed=args.seed)
    fh = eng.decode(zh, 80).cpu().numpy()[0, :T].reshape(4 * T, 20)
    cnr = 20 * np.log10(Hn[:T * 80]) + args.CNRdB
    snr = np.maximum(cnr - 12, 0) + 12 + Gfm - np.maximum(-(cnr - 12), 0) * (1 + Gfm / 3)      # bbfm.py:178-179
    zhn = zh.cpu().numpy()[0, :T].ravel()
    print(f"SNRdB Measured: {10 * np.log10(np.mean(zhn ** 2) / np.mean(10 ** (-snr / 10))):6.2f}")
    out = np.zeros((4 * T, 36), np.float32); out[:, :20] = fh
    if args.features_hat != "/dev/null":
        out.tofile(args.features_hat)
    loss = distortion_loss(feats[:, :20], fh)
    print(f"loss: {loss:5.3f}")
    if args.loss_test > 0.0:
        print("PASS" if loss < args.loss_test else "FAIL")
    if args.write_latent:
        zhn.astype(np.float32).tofile(args.write_latent)
    if args.write_CNRdB:
        cnr.astype(np.float32).tofile(args.write_CNRdB)
    eng.close()
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    cmds = {"txe": _txe, "rxe": _rxe, "inference": _inference, "multipath_samples": _multipath_samples, "bbfm_inference": _bbfm_inference}
    if not argv or argv[0] not in cmds:
        print(__doc__, file=sys.stderr)
        return 2
    return cmds[argv[0]](argv[1:])


if __name__ == "__main__":
    raise SystemExit(main())
