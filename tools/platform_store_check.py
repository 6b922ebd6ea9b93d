#!/usr/bin/env python3
"""developer aid: is a lost / stale 128-byte line a property of THIS library or of the platform when several processes share the GPU?  Plain torch only:
each iteration writes a deterministic pattern into a FRESH buffer with one elementwise kernel on a side stream and compares it with the first result.
Run several instances at once (tools/_gpu_s2.sh); prints the number of differing 128-byte lines per iteration that had any."""
import sys, torch
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda"); st = torch.cuda.Stream()
x = torch.arange(200 * 19200, device=dev, dtype=torch.float32)
bad = []
with torch.cuda.stream(st):
    ref = torch.sin(x * 1e-3) * 0.5
    keep = []
    for it in range(n_iter):
        y = torch.empty_like(x); torch.sin(x * 1e-3, out=y); y.mul_(0.5)
        z = torch.tanh(y) + y                     # a second kernel reading what the first wrote
        d = (y != ref)
        if bool(d.any()):
            idx = d.nonzero().flatten().cpu().numpy()
            bad.append((it, len(idx), sorted(set((idx // 32).tolist()))[:4]))
        keep.append(z if it % 7 == 0 else None)  # vary the allocator's reuse pattern
        if len(keep) > 5: keep.pop(0)
st.synchronize()
print("iterations", n_iter, "with mismatches:", len(bad), bad[:5])
