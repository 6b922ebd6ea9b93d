#!/bin/bash
# developer aid, runs on the GPU box: stall / instruction-cache / LDS counters of the receiver kernel alone (tools/rx_only.py) at one and two
# workgroups per CU (B = 256 / 512).  usage: rx2_counters.sh [variant]
# a rocprofv3 run that aborts can hang until the box's limit (round 5: 30 GPU-minutes lost on an unknown counter name): every run is bounded
rocprofv3() { timeout -k 10 ${RP_TIMEOUT:-420} "$(which rocprofv3)" "$@"; }
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/rxcnt; rm -rf $O; mkdir -p $O; V=${1:-2}
pass() { n=$1; B=$2; shift; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/${n}_$B -o pmc -- python $R/tools/rx_only.py 2 $V $B > $O/${n}_$B.log 2>&1; }
for B in 256 512; do
  pass ic $B SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass wt $B SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
  pass ld $B SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQC_TC_STALL SQC_TC_INST_REQ
  pass in $B SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH
done
python - <<PY
import csv, glob, json, collections
out = {}
for B in (256, 512):
    acc = collections.defaultdict(float); nd = 0
    for n in ("ic", "wt", "ld", "in"):
        f = glob.glob("$O/%s_%d/**/*counter_collection.csv" % (n, B), recursive=True)
        if not f: continue
        rows = [r for r in csv.DictReader(open(f[0])) if "k_rx_sync" in r["Kernel_Name"]]
        names = set(r["Counter_Name"] for r in rows)
        for c in names:
            v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == c]
            acc[c] = sum(v) / len(v)
    out[B] = dict(acc)
    print(B, open("$O/ic_%d.log" % B).read().strip().split("\n")[0][:120])
json.dump(out, open("$O/counters.json", "w"), indent=1)
for c in sorted(out[256]):
    a, b = out[256][c], out[512].get(c, 0.0)
    print(f"{c:32s} {a/1e6:12.2f}M {b/1e6:12.2f}M  x{b/max(a,1):.2f}")
PY
