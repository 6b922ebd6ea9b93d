#!/bin/bash
# developer aid, runs on the GPU box: the default bench command with each of the given libraries in turn, N rounds (A B A B ...), so that
# two builds are compared on one box in one session (boxes differ by +-3 % in sustained clock).  usage: tools/ab_bench.sh N lib1.so lib2.so ... [-- bench args]
N=$1; shift
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 $N); do for L in "${LIBS[@]}"; do
  RADE_LIBRADEHIP=$R/$L python $R/bench.py --no-cpu-baseline --no-roofline --no-parity --steps 60 "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$L', 'round $r', round(d['value']/1e6,2), 'M frames/s', round(d['ms_per_step'],3), 'ms')"
done; done
