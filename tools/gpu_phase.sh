#!/bin/bash
# developer aid (gpurun): build with -DRD_PHASE_TIMING, print the per-phase cycle totals of stream 0 on the bench workload, rebuild the shipped library
cd ${GRAFT_REPO_ROOT:-/root/repo}
( make -C radae_amd/csrc clean >/dev/null; make -C radae_amd/csrc -s EXTRA=-DRD_PHASE_TIMING 2>/dev/null; python tools/phase_timing.py ) 2>&1 | grep -v "amdgpu.ids" | cut -c1-64
make -C radae_amd/csrc clean >/dev/null; make -C radae_amd/csrc -s 2>/dev/null
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['per_class_ms_per_step'])"
