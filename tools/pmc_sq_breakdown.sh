#!/bin/bash
# Where do k_rx_sync's wave-cycles go?  Two PMC passes over one bench step (SQ counters are in quad-cycles, see MI355X_MICROARCH.md).
# a rocprofv3 run that aborts can hang until the box's limit (round 5: 30 GPU-minutes lost on an unknown counter name): every run is bounded
rocprofv3() { timeout -k 10 ${RP_TIMEOUT:-420} "$(which rocprofv3)" "$@"; }
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq; rm -rf $O; mkdir -p $O
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p1 -o pmc -- $CMD > /dev/null 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $O/p2 -o pmc -- $CMD > /dev/null 2> $O/p2.err
python - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    fs = glob.glob("$O/%s/**/pmc_counter_collection.csv" % p, recursive=True)
    if not fs: print(p, "no output", open("$O/%s.err" % p).read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:28]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in acc:
        if "rx_sync" in k or "gemm16<3" in k or "gru_scan" in k:
            print(p, k, {c: "%.3g" % v for c, v in acc[k].items()})
PY
