#!/bin/bash
# build revision $1 (default HEAD) of the library into ab/base.so (git-ignored, travels with gpurun) for tools/ab_cycles.sh
REV=${1:-HEAD}; R=$(cd $(dirname $0)/.. && pwd); T=$(mktemp -d)
git -C $R archive $REV radae_amd/csrc include | tar -x -C $T && make -C $T/radae_amd/csrc -s 2>/dev/null >/dev/null && mkdir -p $R/ab && cp $T/radae_amd/libradehip.so $R/ab/base.so && echo "ab/base.so <- $REV"; rm -rf $T
