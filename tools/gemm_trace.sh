#!/bin/bash
# developer aid, runs on the GPU box: per-dispatch durations of one step's kernels in launch order (one batch in flight) for library $1
# a rocprofv3 run that aborts can hang until the box's limit (round 5: 30 GPU-minutes lost on an unknown counter name): every run is bounded
rocprofv3() { timeout -k 10 ${RP_TIMEOUT:-420} "$(which rocprofv3)" "$@"; }
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/gtrace; rm -rf $O; mkdir -p $O
RADE_LIBRADEHIP=$R/$1 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/bench.py --pipeline 1 --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-roofline > /dev/null 2> $O/err.txt
python - <<PY
import csv, glob
f = glob.glob("$O/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last complete step: from the last k_enc_pack to the following k_rx_sync2
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_enc_pack")]
i0 = idx[-1]
for r in rows[i0:]:
    n = r["Kernel_Name"].split("(")[0][:40]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{n:40s} {d:9.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
    if n.startswith("k_rx_sync2"): break
PY
