#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace stats + three separate PMC passes of the bench command.
# Outputs land in gpurun_out/prof_r01/ and are summarised into profiles/ by tools/make_profile_summary.py.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r01; rm -rf $O; mkdir -p $O
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $CMD > $O/bench_under_rocprof.json 2> $O/stats.err
CMD1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pmc_mfma -o pmc -- $CMD1 > /dev/null 2> $O/pmc_mfma.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- $CMD1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- $CMD1 > /dev/null 2> $O/pmc_write.err
find $O -name "*.csv" | head -20
tail -1 $O/bench_under_rocprof.json | cut -c1-300
