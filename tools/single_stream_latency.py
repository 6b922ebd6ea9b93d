"""BASELINE configs[1]: one stream through the reference's single-stream C ABI (rade_tx / rade_rx, 120 ms modem frame per
call) on one MI355X, and the same 84 modem frames through the plain-C oracle on one host core.  Latency-bound by
construction: reports per-call latency and frames/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radae_amd.api import radae_tx, radae_rx
from radae_amd.channel_tools import synth_features
T = 1008; n_mf = T // 12
feats = synth_features(2024, T)
tx = radae_tx(); rx = radae_rx()
iq = np.zeros((n_mf, 960), np.complex64); out = np.zeros(432, np.float32)
for rep in range(2):                       # first pass warms the device up
    lat_tx = []
    for k in range(n_mf):
        t = time.perf_counter(); tx.do_radae_tx(feats[12 * k:12 * k + 12].ravel(), iq[k]); lat_tx.append(time.perf_counter() - t)
sig = np.concatenate([np.zeros(400, np.complex64), iq.ravel(), np.zeros(2000, np.complex64)])
lat_rx, pos, nvalid = [], 0, 0
while pos + rx.get_nin() <= len(sig):
    nin = rx.get_nin()
    t = time.perf_counter(); r = rx.do_radae_rx(sig[pos:pos + nin], out); lat_rx.append(time.perf_counter() - t)
    pos += nin; nvalid += r & 1
lt, lr = np.array(lat_tx) * 1e3, np.array(lat_rx[3:]) * 1e3
print(f"GPU single stream: rade_tx {lt.mean():.3f} ms/call (p99 {np.percentile(lt, 99):.3f}), rade_rx {lr.mean():.3f} ms/call (p99 {np.percentile(lr, 99):.3f}), "
      f"{nvalid} frames decoded; tx+rx = {12 / ((lt.mean() + lr.mean()) * 1e-3):.0f} feature frames/s = {120.0 / (lt.mean() + lr.mean()):.0f}x real time")
from oracle import oracle_py as O
O.build(); m = O.Model(); otx = O.Tx(m)
t = time.perf_counter(); osig = np.concatenate([otx.frame(feats[12 * k:12 * k + 12].ravel())[0] for k in range(n_mf)]); t_tx = time.perf_counter() - t
full = np.concatenate([np.zeros(400, np.complex64), osig, np.zeros(2000, np.complex64)])
t = time.perf_counter(); d = O.run_rx_stream(m, full); t_rx = time.perf_counter() - t
print(f"CPU oracle, 1 core:  tx {1e3 * t_tx / n_mf:.3f} ms/frame, rx {1e3 * t_rx / len(d['ret']):.3f} ms/call; tx+rx = {T / (t_tx + t_rx):.0f} feature frames/s")
