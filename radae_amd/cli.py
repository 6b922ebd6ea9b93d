"""stdin / stdout filters with the command lines of the reference's Python tools:

    python -m radae_amd.cli txe [--txbpf] [--bypass_enc] [--eoo_data_test] [--model_name BLOB]     features.f32 (or z.f32) -> IQ.f32      /root/reference/radae_txe.py:146-180
    python -m radae_amd.cli rxe [--bypass_dec] [--disable_unsync S] [--foff_err HZ] [--eoo_data_test] [--no_stdout] [-v N] [--model_name BLOB]
                                                                                                     IQ.f32 -> features.f32 (or z_hat.f32)  /root/reference/radae_rxe.py:332-371

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


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in ("txe", "rxe"):
        print(__doc__, file=sys.stderr)
        return 2
    return _txe(argv[1:]) if argv[0] == "txe" else _rxe(argv[1:])


if __name__ == "__main__":
    raise SystemExit(main())
