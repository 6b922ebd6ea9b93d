#!/bin/bash
# developer aid (gpurun): A/B of two builds of the library on the per-stream cycle counts of the receiver launch (mean over streams is
# free of the tail noise a kernel duration carries).  ab/base.so = the build to compare against (tools/ab_build_base.sh <rev>).
cd ${GRAFT_REPO_ROOT:-/root/repo}
summ() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['seeds']
print('$1', ' '.join(f\"seed{k}: mean {v['cycles_mean']/1e6:.3f}M max {v['cycles_max']/1e6:.3f}M ms {v['kernel_ms']:.3f} search {v['cycles_per_search_call_fit']/1e3:.1f}k sync {v['cycles_per_sync_call_fit']/1e3:.1f}k |\" for k,v in d.items()))"; }
python tools/stream_cycles.py 2>/dev/null | summ new
if [ -f ab/base.so ]; then
  cp radae_amd/libradehip.so /tmp/new.so; cp ab/base.so radae_amd/libradehip.so
  python tools/stream_cycles.py 2>/dev/null | summ base
  cp /tmp/new.so radae_amd/libradehip.so
fi
