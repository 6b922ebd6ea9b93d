#!/bin/bash
# developer aid, runs on the GPU box: the default bench with and without an environment switch, N rounds (A B A B ...).  usage: tools/ab_env.sh N VAR=value [-- bench args]
N=$1; shift; KV=$1; shift; [ "$1" == "--" ] && shift
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python $R/bench.py --no-cpu-baseline --no-roofline --no-parity --steps 60 "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,2), 'M frames/s', round(d['ms_per_step'],3), 'ms')"; }
for r in $(seq 1 $N); do echo "default   round $r $(one "$@")"; echo "$KV round $r $(env $KV bash -c "$(declare -f one); R=$R one $*")"; done
