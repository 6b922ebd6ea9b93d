#!/bin/bash
# developer aid, runs on the GPU box: the interleaved A/B of libraries on the default bench -- a wrapper of tools/ab_bench.py (median, MAD, paired differences,
# decision rule |median difference| >= 2 x MAD; >= 10 rounds).  usage: tools/ab_bench.sh N lib1.so lib2.so ... [-- bench args]   (the first library is the baseline)
N=$1; shift
V=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do V+=("$(basename ${1%.so})=$1"); shift; done
R=${GRAFT_REPO_ROOT:-/root/repo}
exec python $R/tools/ab_bench.py --rounds $N "${V[@]}" "$@"
