#!/bin/bash
# developer aid, runs on the GPU box: which phases of k_rx_sync2 overlap with the other workgroup on their CU?  For every census mask (tools/rx2_census.sh) the
# kernel's duration is measured with 256 streams (one workgroup per CU) and 512 (two): a phase that runs twice costs d1 = t(mask) - t(0) alone and d2 beside a
# second workgroup; d2 / d1 = 1 means the phase's time is fully hidden behind / beside the neighbour's work, 2 means the two workgroups serialise on it.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pair; rm -rf $O; mkdir -p $O
export RADE_LIBRADEHIP=$R/abso/census.so
for B in 256 512; do for m in 0 1 2 4 16 32 128 256 512; do
  RADE_RX2_CENSUS=$m python $R/tools/rx_only.py 6 2 $B 2>/dev/null | tail -1 > $O/b${B}_m$m.txt
done; done
python - <<PY
import re, json
names = {1: "decoder stage (skipped)", 2: "GRU recurrence (skipped)", 4: "FIR", 16: "refine", 32: "check_pilots rows", 128: "corrected window", 256: "demodulator DFT", 512: "pilot search"}
t = {}
for B in (256, 512):
    for m in (0, 1, 2, 4, 16, 32, 128, 256, 512):
        s = open("$O/b%d_m%d.txt" % (B, m)).read()
        t[(B, m)] = float(re.search(r"ms/launch ([0-9.]+)", s).group(1))
out = {"ms_per_launch": {f"{B}_{m}": v for (B, m), v in t.items()}, "phases": {}}
print("baseline ms/launch: 256 streams %.3f, 512 streams %.3f (x%.2f)" % (t[(256, 0)], t[(512, 0)], t[(512, 0)] / t[(256, 0)]))
for m, n in names.items():
    d1, d2 = abs(t[(256, m)] - t[(256, 0)]), abs(t[(512, m)] - t[(512, 0)])
    out["phases"][n] = {"alone_ms": d1, "paired_ms": d2, "ratio": d2 / d1 if d1 else None}
    print(f"{n:28s} alone {d1:6.3f} ms   paired {d2:6.3f} ms   ratio {d2 / max(d1, 1e-9):5.2f}")
json.dump(out, open("$O/pair_overlap.json", "w"), indent=1)
PY
