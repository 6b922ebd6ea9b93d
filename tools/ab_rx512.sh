#!/bin/bash
# developer aid, runs on the GPU box: the receiver kernel alone at two workgroups per CU (tools/rx_only.py, 512 streams) for each library in turn, N rounds
N=$1; shift; R=${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 $N); do for L in "$@"; do
  echo "$L round $r $(RADE_LIBRADEHIP=$R/$L python $R/tools/rx_only.py 8 2 512 2>/dev/null | tail -1)"
done; done
