#!/bin/bash
# developer aid, runs on the GPU box: mean per-stream cycles of the receiver launch (tools/stream_cycles.py) for each library in turn, N rounds
N=$1; shift; R=${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 $N); do for L in "$@"; do
  RADE_LIBRADEHIP=$R/$L python $R/tools/stream_cycles.py 2>/dev/null | python -c "
import json,sys
c=json.load(sys.stdin); v=c['seeds']['1']
print('$L', 'round $r', 'mean cycles', round(v['cycles_mean']), 'search call', round(v['cycles_per_search_call_fit']), 'sync call', round(v['cycles_per_sync_call_fit']), 'ms', round(v['kernel_ms'],3))"
done; done
