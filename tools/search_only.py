"""Per-call cost of the search state: noise-only input keeps every stream searching (cached |Dt| surface)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radae_amd.engine import BatchEngine
B, ncalls = 256, 60
eng = BatchEngine(B, max_tx_mf=1)
dev = torch.device('cuda')
g = torch.Generator(device=dev); g.manual_seed(1)
rx = (0.05 * torch.randn((B, 960 * ncalls, 2), device=dev, generator=g)).contiguous()
rx = torch.view_as_complex(rx)
for rep in range(2):
    eng.rx_reset()
    eng.profile(True)
    fo, st, _ = eng.rx(rx); torch.cuda.synchronize()
    eng.profile(False); pr = eng.profile_get()
    print("calls", st[0].n_calls, "states", set(s.state for s in st), {k: (round(v["ms"], 3), v["launches"]) for k, v in pr.items() if v["launches"]},
          "us per call", 1e3 * pr["rx_sync"]["ms"] / max(st[0].n_calls, 1))
try:
    import ctypes as C
    from radae_amd.engine import load_library
    lib = load_library(); lib.rd_debug_phase_cycles.argtypes = [C.c_void_p]
    buf = (C.c_longlong * 24)(); lib.rd_debug_phase_cycles(buf)
    names = ["load", "bpf+shift", "detect corr", "detect reduce", "refine", "check rows", "sigma+corr+slip", "freqcorr", "demod dft", "eq", "statemachine", "store", "r:tables", "r:mfma", "r:scan", "r:argmax", "detect:pre", "detect:fft"]
    for i, n in enumerate(names):
        if buf[i]: print(f"{n:16s} {buf[i] / (2 * ncalls):10.0f} cycles per call")
except AttributeError:
    pass
