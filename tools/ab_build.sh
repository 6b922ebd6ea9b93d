#!/bin/bash
# developer aid: build the WORKING TREE's library with extra hipcc flags into gpu_ab/<name>.so (git-ignored, travels with gpurun):
#   tools/ab_build.sh timing -DRD_PHASE_TIMING ; on the GPU box: RADE_LIBRADEHIP=gpu_ab/timing.so python tools/phase_timing.py
NAME=${1:?name}; shift; R=$(cd $(dirname $0)/.. && pwd); T=$(mktemp -d)
mkdir -p $T/radae_amd $R/gpu_ab && cp -r $R/radae_amd/csrc $T/radae_amd/ && cp -r $R/include $T/ && rm -f $T/radae_amd/csrc/*.o
make -C $T/radae_amd/csrc -s EXTRA="$*" 2>/dev/null >/dev/null && cp $T/radae_amd/libradehip.so $R/gpu_ab/$NAME.so && echo "gpu_ab/$NAME.so <- working tree, EXTRA=$*"; rm -rf $T
