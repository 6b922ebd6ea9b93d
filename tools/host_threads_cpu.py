#!/usr/bin/env python3
"""developer aid: where the HOST's CPU time of the pipelined bench loop goes, per thread (utime + stime of every task of the process from /proc/self/task, with its name):
the three lane threads (launches + waits), the interpreter's main thread, and whatever threads the HIP runtime keeps.  usage: [RADE_SYNC=block|spin] [taskset -c 0,1] host_threads_cpu.py [steps]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, sigma_from_EbNodB
from radae_amd.channel_tools import synth_features, multipath_g
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 900
B, T, depth = 256, 1008, 3; n_mf = T // 12
dev = torch.device("cuda")
feats = torch.tensor(np.stack([synth_features(1000 + b, T) for b in range(B)]), device=dev)
G = torch.tensor(np.stack([multipath_g("mpp", 8000, n_mf * 960, 5000 + b) for b in range(B)]), device=dev)
engs = [BatchEngine(B, max_tx_mf=n_mf) for _ in range(depth)]
lanes = [torch.cuda.Stream(device=dev) for _ in range(depth)]
outs = [(torch.zeros((B, 120, 432), dtype=torch.float32, device=dev), torch.zeros((B, 180), dtype=torch.float32, device=dev)) for _ in range(depth)]
tids = {}
def tasks():
    r = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read().rsplit(")", 1)[1].split()
            r[int(t)] = (open(f"/proc/self/task/{t}/comm").read().strip(), (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK"))
        except Exception:
            pass
    return r
lane_cpu = [0.0] * depth; lane_parts = [dict(reset=0.0, tx_channel=0.0, rx=0.0) for _ in range(depth)]
def lane(i, n):
    tids[threading.get_native_id()] = f"lane {i}"
    c0 = time.thread_time()
    with torch.cuda.stream(lanes[i]):
        for k in range(i, n, depth):
            e = engs[i]; p = lane_parts[i]
            a = time.thread_time(); e.reset(); b_ = time.thread_time(); p["reset"] += b_ - a
            rx = e.tx_channel(feats, sigma_from_EbNodB(3.0), -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=1 + k); c_ = time.thread_time(); p["tx_channel"] += c_ - b_
            e.rx(rx, features_out=outs[i][0], eoo_out=outs[i][1]); p["rx"] += time.thread_time() - c_
    lane_cpu[i] = time.thread_time() - c0
def run(n):
    ths = [threading.Thread(target=lane, args=(i, n)) for i in range(depth)]
    [t.start() for t in ths]; [t.join() for t in ths]
run(24); torch.cuda.synchronize()
lane_parts = [dict(reset=0.0, tx_channel=0.0, rx=0.0) for _ in range(depth)]
t0 = tasks(); w0 = time.perf_counter(); p0 = time.process_time()
run(NS); torch.cuda.synchronize()
w = time.perf_counter() - w0; t1 = tasks()
tids[threading.get_native_id()] = "main"
rows = sorted(((t1[t][1] - t0.get(t, (None, 0.0))[1], t1[t][0], t) for t in t1), reverse=True)
print(f"{NS} steps in {w:.3f} s = {1e3 * w / NS:.3f} ms per step, {B * T * NS / w / 1e6:.1f} M frames/s; RADE_SYNC={os.environ.get('RADE_SYNC', 'auto')} cpus={len(os.sched_getaffinity(0))}")
print(f"CPU seconds per step by thread (all {len(t1)} tasks; lane threads have ended and are not listed):")
for dt, name, t in rows[:12]:
    print(f"  {1e3 * dt / NS:8.3f} ms  {name:20s} {tids.get(t, '')}")
print(f"  lane threads (thread_time): {[round(1e3 * c / NS, 3) for c in lane_cpu]} ms per step each; by call, summed over lanes: " + str({k: round(1e3 * sum(p[k] for p in lane_parts) / NS, 3) for k in lane_parts[0]}))
print(f"  process CPU {1e3 * (time.process_time() - p0) / NS:.3f} ms per step = {(time.process_time() - p0) / w:.2f} cores busy")
