#!/bin/bash
# developer aid (runs here, hipcc cross-compiles): build the library variants whose keep / reject calls round 4 took inside the noise, for a re-run under
# tools/ab_bench.py:   abso/base.so (HEAD)   abso/noprio.so (no s_setprio(3) on the scans / one-thread-per-stream kernels)
#                      abso/coal.so (tools/experiments/gemm16c_coalesced.inc dispatched)   abso/scangin.so (tools/experiments/gru_scan_gin.patch)
set -e
R=$(cd $(dirname $0)/.. && pwd); mkdir -p $R/abso
build() { # name, tree
  make -C $2/radae_amd/csrc -s EXTRA="$3" >/dev/null 2>&1 && cp $2/radae_amd/libradehip.so $R/abso/$1.so && echo "abso/$1.so built"
}
fresh() { T=$(mktemp -d); git -C $R archive HEAD radae_amd/csrc include tools/experiments | tar -x -C $T; echo $T; }
T=$(fresh); build base $T ""; rm -rf $T
T=$(fresh); sed -i 's/__builtin_amdgcn_s_setprio(3);/;/' $T/radae_amd/csrc/rade_kernels.hip $T/radae_amd/csrc/rade_rx.hip; build noprio $T ""; rm -rf $T
T=$(fresh)
python3 - $T <<'PY'
import sys
T = sys.argv[1]
inc = open(T + "/tools/experiments/gemm16c_coalesced.inc").read()
kern, disp = inc.split("// ---- dispatch (inside rd_launch_gemm, ahead of the k_gemm16p<3, 1, true> branch) ----")
p = T + "/radae_amd/csrc/rade_kernels.hip"; s = open(p).read()
s = s.replace('extern "C" int rd_launch_gemm(', kern + '\nextern "C" int rd_launch_gemm(', 1)
anchor = "            if (a->Wscale) { dim3 g1(gx, ntt / 3); hipLaunchKernelGGL((k_gemm16p<3, 1, true>)"
assert anchor in s
s = s.replace(anchor, disp + "\n" + anchor, 1)
open(p, "w").write(s)
PY
build coal $T ""; rm -rf $T
T=$(fresh); (cd $T && patch -p1 -s < tools/experiments/gru_scan_gin.patch); build scangin $T ""; rm -rf $T
ls -la $R/abso/
