import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine, load_library, sigma_from_EbNodB
from radae_amd.channel_tools import synth_features, multipath_g
B, T = int(os.environ.get("PT_STREAMS", "256")), 1008; n_mf = T // 12
eng = BatchEngine(B, max_tx_mf=n_mf, rx_trace_calls=int(os.environ.get("PT_TRACE", "128")))
dev = torch.device('cuda')
feats = torch.tensor(np.stack([synth_features(1000 + b, T) for b in range(B)]), device=dev)
iq = eng.tx(feats)
G = torch.empty((B, n_mf * 960, 2), dtype=torch.complex64, device=dev)
for b in range(B):
    G[b] = torch.from_numpy(multipath_g("mpp", 8000, n_mf * 960, 5000 + b)).to(dev)
rx = eng.channel(iq, sigma_from_EbNodB(3.0), -11.0, n_pre=8000, n_post=1152, with_eoo=True, G=G, seed=1)
lib = load_library(); lib.rd_debug_phase_cycles2.argtypes = [C.c_void_p]
buf = (C.c_longlong * 32)(); lib.rd_debug_phase_cycles2(buf)
eng.profile(True)
fo, st, _ = eng.rx(rx); torch.cuda.synchronize()
eng.profile(False); pr = eng.profile_get()['rx_sync']
lib.rd_debug_phase_cycles2(buf)
names = ["load", "loop top", "decode+post (total)", "input: filtered samples requested + pilot replicas", "(unused)", "rx_buf shift + append", "detect: pre", "detect: correlator", " check: two-stage correlator (wave 0)", "planes+refine", "check: barrier wait", " check: window share (wave 0)", "state machine + dft", "search: reduce / sync: eq", "call end", " search: stage A(0)", " search: k loop (per group)", " search: epilogue", " search: end barrier", " dec: rx_buf + row sums out to HBM and back (both ways)", " dec: hist+dense1+gin0", " dec: scan", " dec: glu", " dec: conv+next", " dec: fixup", " dft: prefetch + products (wave 0)", " dft: barrier wait", " refine: window to f64", " refine: moments (f64 mfma)", " refine: sum partials", " refine: polynomials", " refine: argmax"]
tot = sum(buf[:15])
print("stream0 calls", st[0].n_calls, "valid", st[0].n_valid, "rx_sync kernel ms", pr["ms"], "launches", pr["launches"])
for i, n in enumerate(names): print(f"{n:18s} {buf[i]:12d} cyc  {100*buf[i]/tot:5.1f}%  {buf[i]/100e6*1e3:8.3f} ms (100MHz clk?)")

print("total cycles block0", tot, "=> if block0 were busy the whole time: clock >=", tot/ (pr["ms"]*1e-3)/1e9, "GHz")

t = eng.rx_trace(0); sb = t["state_before"]
print("stream0 call mix: search", int((sb == 0).sum()), "candidate", int((sb == 1).sum()), "sync", int((sb == 2).sum()))
