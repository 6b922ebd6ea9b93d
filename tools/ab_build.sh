#!/bin/bash
# developer aid: build the WORKING TREE's library with extra hipcc flags into abso/<name>.so (git-ignored, travels with gpurun):
#   tools/ab_build.sh timing -DRD_PHASE_TIMING ; on the GPU box: RADE_LIBRADEHIP=abso/timing.so python tools/phase_timing.py
NAME=${1:?name}; shift; R=$(cd $(dirname $0)/.. && pwd); T=$(mktemp -d)
mkdir -p $T/radae_amd $R/abso && cp -r $R/radae_amd/csrc $T/radae_amd/ && cp -r $R/include $T/ && rm -f $T/radae_amd/csrc/*.o
make -C $T/radae_amd/csrc -s EXTRA="$*" 2>/dev/null >/dev/null && cp $T/radae_amd/libradehip.so $R/abso/$NAME.so && echo "abso/$NAME.so <- working tree, EXTRA=$*"; rm -rf $T
