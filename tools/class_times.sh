#!/bin/bash
# per-class kernel times (one batch alone) of one or more builds: tools/class_times.sh lib.so [lib.so ...]
for lib in "$@"; do
  echo "== $lib"
  RADE_LIBRADEHIP=$lib python bench.py --pipeline 1 --no-cpu-baseline --no-parity --steps 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print(r.get('per_class_ms_per_step'), r.get('sum_kernel_ms_per_step'))"
done
