"""Throughput of BASELINE configs 1 (model05, rate-Rs AWGN) and 5 (BBFM, FM-demodulator channel) with 256 streams x 1008
feature frames on one MI355X: encode -> symbol-rate channel (on-chip Philox noise) -> decode, inputs resident in HBM."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, DEFAULT_BLOB
B, T = 256, 1008
results = []
W = os.path.dirname(DEFAULT_BLOB)
for name, blob, mode, p0, p1 in (("config 1 model05 rate-Rs AWGN 10 dB", "model05.bin", "rs", 10 ** (-10 / 20), 0.0), ("config 5 bbfm CNR 20 dB", "bbfm_random_seed20240501.bin", "bbfm", 20.0, 13.47)):
    eng = BatchEngine(B, max_tx_mf=T // 12, blob=os.path.join(W, blob), flags=0x100)
    feats = torch.randn((B, T // 4, 80), device="cuda") * 0.5
    def step(k):
        eng.reset()
        z = eng.encode(feats)
        zh = eng.channel_symbol(z, mode, p0, p1, seed=10 + k)
        return eng.decode(zh, 80)
    for k in range(2): step(k)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
    for k in range(n): out = step(k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name}: {1e3 * dt:.2f} ms per batch, {B * T / dt / 1e6:.1f} M feature frames/s")
    results.append({"config": name, "blob": blob, "streams": B, "frames_per_stream": T, "ms_per_batch": 1e3 * dt, "frames_per_s": B * T / dt, "steps_timed": n,
                    "path": "rade_batch_encode -> rade_batch_channel_symbol (device Philox noise) -> rade_batch_decode, inputs resident in HBM"})
    eng.close()
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "config_rates.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(results, open(out, "w"), indent=1)
