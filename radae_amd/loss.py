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
