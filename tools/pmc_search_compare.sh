cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -oE "\bSQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $R/gpurun_out/sq_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  RADE_ROUND_CALLS=128 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcA_$n -- python $R/tools/search_only.py > /dev/null 2>&1
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcB_$n -- $R/tools/ubench/detect_bench > /dev/null 2>&1
done
cd $R; for f in $(find gpurun_out/pmcA_* gpurun_out/pmcB_* -name "*counter_collection.csv"); do python tools/pmc_summary.py $f | grep -E "k_rx_sync|k_detect_bench|^#"; done
