#!/bin/bash
# developer aid, runs on the GPU box: where the LDS bank conflicts of k_rx_sync2 sit -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the receiver alone (tools/rx_only.py)
# for census masks of a -DRX2_CENSUS build (tools/ab_build.sh census -DRX2_CENSUS): a phase run TWICE (mask 8 planes, 16 refine, 32 check_pilots rows + window, 256 DFT,
# 512 pilot search) adds its own conflict and active cycles to the totals; mask 1 runs no decoder stage.
# a rocprofv3 run that aborts can hang until the box's limit (round 5: 30 GPU-minutes lost on an unknown counter name): every run is bounded
rocprofv3() { timeout -k 10 ${RP_TIMEOUT:-420} "$(which rocprofv3)" "$@"; }
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ldsconf; rm -rf $O; mkdir -p $O
export RADE_LIBRADEHIP=$R/abso/census.so
for m in 0 1 8 16 32 256 512; do
  export RADE_RX2_CENSUS=$m
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/m$m -o pmc -- python $R/tools/rx_only.py 2 2 > $O/m$m.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out = {}
for m in (0, 1, 8, 16, 32, 256, 512):
    f = glob.glob("$O/m%d/**/*counter_collection.csv" % m, recursive=True)
    acc = collections.defaultdict(float); nd = 0
    for row in csv.DictReader(open(f[0])):
        if "k_rx_sync2" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); nd += row["Counter_Name"] == "SQ_LDS_IDX_ACTIVE"
    out[m] = {k: v / max(nd, 1) for k, v in acc.items()}
b = out[0]
rep = {"per_launch_of_512_streams": out, "share_of_lds_cycles_that_are_conflicts": b["SQ_LDS_BANK_CONFLICT"] / b["SQ_LDS_IDX_ACTIVE"], "phases": {}}
names = {1: "decoder stage (mask 1: removed)", 8: "operand planes", 16: "refine()", 32: "check_pilots rows + correlations + window", 256: "demodulator DFT", 512: "pilot search"}
for m, n in names.items():
    sgn = -1.0 if m == 1 else 1.0
    dc = sgn * (out[m]["SQ_LDS_BANK_CONFLICT"] - b["SQ_LDS_BANK_CONFLICT"]); da = sgn * (out[m]["SQ_LDS_IDX_ACTIVE"] - b["SQ_LDS_IDX_ACTIVE"])
    rep["phases"][n] = {"conflict_cycles": dc, "lds_active_cycles": da, "share_of_all_conflicts": dc / b["SQ_LDS_BANK_CONFLICT"], "conflict_share_inside_phase": dc / da if da else None}
json.dump(rep, open("$O/lds_conflicts.json", "w"), indent=1)
print(json.dumps(rep["phases"], indent=1)); print("whole kernel:", rep["share_of_lds_cycles_that_are_conflicts"])
PY
