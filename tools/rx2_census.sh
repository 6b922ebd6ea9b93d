#!/bin/bash
# developer aid, runs on the GPU box: instruction counters of k_rx_sync2 alone (tools/rx_only.py) for each census mask of a -DRX2_CENSUS build
# (tools/ab_build.sh census -DRX2_CENSUS).  Mask bits: 1 no decoder stage, 2 no GRU recurrence, 8 operand planes twice,
# 16 refine twice, 32 check_pilots rows twice, 64 correlations twice, 128 corrected window twice, 256 demodulator DFT twice, 512 detect_pilots (matrix-core pilot search) twice.
# a rocprofv3 run that aborts can hang until the box's limit (round 5: 30 GPU-minutes lost on an unknown counter name): every run is bounded
rocprofv3() { timeout -k 10 ${RP_TIMEOUT:-420} "$(which rocprofv3)" "$@"; }
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/census; rm -rf $O; mkdir -p $O
export RADE_LIBRADEHIP=$R/abso/census.so
for m in 0 1 2 8 16 32 64 128 256 512; do
  export RADE_RX2_CENSUS=$m
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/m$m -o pmc -- python $R/tools/rx_only.py 2 2 > $O/m$m.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out = {}
for m in (0, 1, 2, 8, 16, 32, 64, 128, 256, 512):
    f = glob.glob("$O/m%d/**/*counter_collection.csv" % m, recursive=True)
    acc = collections.defaultdict(float); n = 0
    for row in csv.DictReader(open(f[0])):
        if "k_rx_sync2" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
    nd = sum(1 for row in csv.DictReader(open(f[0])) if "k_rx_sync2" in row["Kernel_Name"] and row["Counter_Name"] == "SQ_INSTS_VALU")
    out[m] = {k: v / max(nd, 1) for k, v in acc.items()}; out[m]["dispatches"] = nd
    out[m]["log"] = open("$O/m%d.log" % m).read().strip().split("\n")[-1]
json.dump(out, open("$O/census.json", "w"), indent=1)
base = out[0]
for m, d in out.items():
    print(m, {k: (round(v / 1e6, 2) if isinstance(v, float) else v) for k, v in d.items() if k != "log"}, d["log"][-60:])
PY
