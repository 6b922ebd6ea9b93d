#!/usr/bin/env python3
"""Turn gpurun_out/prof_r01/ (tools/collect_profiles.sh) into the tracked summaries under profiles/:
r01_bench_kernel_stats.csv, r01_bench_under_rocprof.json, r01_pmc_per_kernel.txt, r01_pmc_summary.json."""
import csv, json, os, shutil, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", "prof_r01"); dst = os.path.join(R, "profiles")
shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, "r01_bench_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_under_rocprof.json")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(dst, "r01_bench_under_rocprof.json"), "w"), indent=1)

def load(name):
    rows = list(csv.DictReader(open(os.path.join(src, name, "pmc_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in cnt.items()}
out = {"command": "rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline (three separate passes: SQ/GRBM set, FETCH_SIZE, WRITE_SIZE; tools/collect_profiles.sh)",
       "note": "FETCH_SIZE/WRITE_SIZE are the raw counters x 1024 (KB units, MI355X_MICROARCH.md); the gfx950 wide-load halving correction is NOT applied (the dominant kernel mixes 4/8/16-byte per-lane loads: uncalibrated width). GRBM_GUI_ACTIVE is summed over the 8 XCDs, so cycles = GRBM/8; MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs) / (cycles x 1024 SIMDs).",
       "kernels": {}}
mf, n1 = load("pmc_mfma"); fe, n2 = load("pmc_fetch"); wr, n3 = load("pmc_write")
txt = []
for name, (agg, n) in (("pmc_mfma", (mf, n1)), ("pmc_fetch", (fe, n2)), ("pmc_write", (wr, n3))):
    txt.append(f"# gpurun_out/prof_r01/{name}/pmc_counter_collection.csv")
    for k in sorted(agg, key=lambda k: -n[k]):
        txt.append(f"{k[:42]:42s} dispatches {n[k]:5d} " + " ".join(f"{c}={v / n[k]:.4g}/disp" for c, v in sorted(agg[k].items())))
open(os.path.join(dst, "r01_pmc_per_kernel.txt"), "w").write("\n".join(txt) + "\n")
for k in mf:
    n = n1[k]; a = mf[k]
    out["kernels"][k] = {
        "dispatches": n, "gpu_cycles_per_dispatch": a.get("GRBM_GUI_ACTIVE", 0.0) / n / 8.0,
        "mfma_busy_pct": 100.0 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0) if a.get("GRBM_GUI_ACTIVE", 0.0) else 0.0,
        "mfma_mops_f32_per_dispatch": a.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / n, "mfma_mops_f64_per_dispatch": a.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / n,
        "mfma_mops_f16_per_dispatch": a.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) / n,
        "fetch_bytes_per_dispatch": 1024.0 * fe.get(k, {}).get("FETCH_SIZE", 0.0) / max(n2.get(k, 1), 1),
        "write_bytes_per_dispatch": 1024.0 * wr.get(k, {}).get("WRITE_SIZE", 0.0) / max(n3.get(k, 1), 1)}
json.dump(out, open(os.path.join(dst, "r01_pmc_summary.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    if v["gpu_cycles_per_dispatch"] > 1e5: print(f"{k[:40]:40s} n={v['dispatches']:4d} cyc={v['gpu_cycles_per_dispatch']:.3g} mfma_busy={v['mfma_busy_pct']:.1f}% fetch={v['fetch_bytes_per_dispatch']/1e6:.1f}MB write={v['write_bytes_per_dispatch']/1e6:.1f}MB")
