#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace stats of the bench command + separate PMC passes (never combined with
# trace domains other than --kernel-trace).  Outputs land in gpurun_out/prof_$TAG/ and are summarised into profiles/ by
# tools/make_profile_summary.py (run it afterwards in the repo).  Usage: bash tools/collect_profiles.sh [tag]
# a rocprofv3 run that aborts can hang until the box's limit (round 5: 30 GPU-minutes lost on an unknown counter name): every run is bounded
rocprofv3() { timeout -k 10 ${RP_TIMEOUT:-420} "$(which rocprofv3)" "$@"; }
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
# 1. the plain bench line (un-profiled) that the roofline block is quoted from
python $R/bench.py --steps 100 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
# 2. rocprofv3 --kernel-trace --stats of the same command (shorter: the trace of 100 steps is large)
#    --pipeline 1: kernel durations with one batch in flight, which is how bench.py's roofline leg times them (HIP events); the
#    second pass is the default command (two batches in flight: a launch's duration then includes CUs shared with the other batch)
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --pipeline 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $CMD > $O/bench_under_rocprof.json 2> $O/stats.err
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pipe -o bench -- $CMD > $O/bench_pipelined_under_rocprof.json 2> $O/stats_pipe.err
# 3. counters, one pass each
CMD1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity"
pass() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o pmc -- $CMD1 > /dev/null 2> $O/pmc_$n.err; }
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F16
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
# 3b. the other configurations and variants (plain lines, no profiler): configs[1] alone, other pipeline depths, the two-call channel, configs 1 and 5 rates,
#     the C host, the single-stream ABI, rocprofv3 stats of the single-stream step kernels
python $R/bench.py --config 2 > $O/bench_config2.json 2> $O/bench_config2.err
for p in 1 2 4; do python $R/bench.py --steps 60 --pipeline $p --no-cpu-baseline --no-parity --no-roofline > $O/bench_p$p.json 2> /dev/null; done
python $R/bench.py --steps 60 --two-pass-channel --no-cpu-baseline --no-parity --no-roofline > $O/bench_two_pass_channel.json 2> /dev/null
python $R/tools/config_rates.py > $O/config_rates.txt 2>&1; cp $R/gpurun_out/config_rates.json $O/ 2>/dev/null
(cd $R && ./hosts/rade_multi_bench --gpus 1 --steps 100 --warmup 420 > $O/c_host_pipeline3.json 2> /dev/null)    # (the warm-up is the 1.5 s of pre-warm bench.py does)
python $R/tools/single_stream_latency.py > $O/single_stream_latency.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o c2 -- python $R/bench.py --config 2 > /dev/null 2> $O/stats_c2.err
# 4. per-stream duration of the receiver launch (tail analysis)
python $R/tools/stream_cycles.py > $O/stream_cycles.json 2> $O/stream_cycles.err
# 5. developer-build runs (built here beforehand: tools/ab_build.sh census -DRX2_CENSUS ; tools/ab_build.sh timing -DRD_PHASE_TIMING): the per-phase
#    instruction census of k_rx_sync2, its stall / LDS / instruction-cache counters at one and two workgroups per CU, and the per-phase cycle table
if [ -f $R/abso/census.so ]; then bash $R/tools/rx2_census.sh > $O/rx2_census.txt 2>&1; cp $R/gpurun_out/census/census.json $O/rx2_census.json 2>/dev/null; fi
bash $R/tools/rx2_counters.sh 2 > $O/rx2_counters.txt 2>&1; cp $R/gpurun_out/rxcnt/counters.json $O/rx2_counters.json 2>/dev/null
cd /tmp
if [ -f $R/abso/timing.so ]; then RADE_LIBRADEHIP=$R/abso/timing.so python $R/tools/phase_timing2.py > $O/phase_timing_rx2.txt 2>/dev/null; fi
find $O -name "*.csv" | head -30
tail -c 400 $O/bench_line.json
