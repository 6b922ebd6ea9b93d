"""Objective metric of the reference: `distortion_loss` (radae/radae_base.py:50-68) and the
time-alignment search of `loss.py:find_loss` (:64-91).  Offline host tool (numpy), as in the reference."""
from __future__ import annotations

import numpy as np


def distortion_loss(y_true: np.ndarray, y_pred: np.ndarray) -> float:
    """y_*: (T, >=20|21) float32; uses 20 features, or 21 when the last dim is exactly 21."""
    d = y_true.shape[-1]
    if d not in (20, 21):
        y_true, y_pred, d = y_true[..., :20], y_pred[..., :20], 20
    yt = y_true.astype(np.float32); yp = y_pred.astype(np.float32)
    ceps = yp[..., :18] - yt[..., :18]
    pitch = np.float32(2.0) * (yp[..., 18:19] - yt[..., 18:19])
    corr = yp[..., 19:20] - yt[..., 19:20]
    pw = np.maximum(yt[..., 19:20] + np.float32(0.5), 0) ** 2
    data = (yp[..., 20:21] - yt[..., 20:21]) if d == 21 else np.float32(0.0)
    per = ceps ** 2 + np.float32(3.0 * (10 / 18)) * np.abs(pitch) * pw + np.float32(1 / 18) * corr ** 2 + np.float32(0.5 / 18) * data ** 2
    return float(np.mean(np.mean(per, axis=-1)))


def find_loss(features: np.ndarray, features_hat: np.ndarray):
    """Slide features_hat over features (both (T, 36) stride-36 files), return (min loss, start)."""
    f = features[:, :20]; h = features_hat[:, :20]
    n, nh = len(f), len(h)
    assert 0 < nh <= n
    best, start = distortion_loss(f[:nh], h), 0
    for s in range(n - nh):
        l = distortion_loss(f[s:s + nh], h)
        if l < best:
            best, start = l, s
    return best, start


def main(argv=None) -> int:
    """The command line of the reference's loss.py (:36-112, without --plot): time-aligned loss between two `.f32` feature files, PASS / FAIL against thresholds;
    what its ctests run after every receive pipeline (`python3 loss.py features_in.f32 features_out.f32 --loss_test 0.15 --acq_time_test 0.5 --clip_end 100`)."""
    import argparse
    ap = argparse.ArgumentParser(prog="radae_amd.loss")
    ap.add_argument("features", type=str, help="path to input feature file in .f32 format")
    ap.add_argument("features_hat", type=str, help="path to output feature file in .f32 format")
    ap.add_argument("--features_hat2", type=str, help="path to optional 2nd features file to compare two runs")
    ap.add_argument("--loss_test", type=float, default=0.0, help="compare loss to arg, print PASS/FAIL")
    ap.add_argument("--acq_time_test", type=float, default=0, help="compare acquisition time to threshold arg, print PASS/FAIL")
    ap.add_argument("--clip_start", type=int, default=0)
    ap.add_argument("--clip_end", type=int, default=0)
    ap.add_argument("--compare", action="store_true", help="compare features_hat and features_hat2")
    args = ap.parse_args(argv)

    def load(fn):
        f = np.fromfile(fn, np.float32)
        return f.reshape(-1, 36)

    def one(fn_hat):
        f, h = load(args.features), load(fn_hat)
        h = h[args.clip_start:len(h) - args.clip_end]
        loss, start = find_loss(f, h)
        print(f"Loss between {args.features:s} and {fn_hat:s}")
        print(f"  loss: {loss:5.3f} start: {start:d} acq_time: {start * 0.01:5.2f} s")
        return loss, start

    loss, start = one(args.features_hat)
    if args.loss_test > 0.0 and loss > args.loss_test:
        print("FAIL"); return 0                            # (the reference prints and quits with status 0; its ctests grep for PASS)
    if args.acq_time_test > 0 and start * 0.01 > args.acq_time_test:
        print("FAIL"); return 0
    if args.loss_test > 0.0 or args.acq_time_test:
        print("PASS")
    if args.features_hat2:
        loss2, _ = one(args.features_hat2)
        if args.compare:
            print(f"loss1: {loss:5.3f} loss2: {loss2:5.3f} delta: {abs(loss - loss2):5.3f}")
            if abs(loss - loss2) < 0.01:
                print("PASS")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
