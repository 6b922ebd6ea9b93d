#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the tracked summaries under profiles/:
<tag>_bench_kernel_stats.csv, <tag>_bench_line.json, <tag>_bench_under_rocprof.json, <tag>_pmc_per_kernel.txt, <tag>_pmc_summary.json,
<tag>_stream_cycles.json.  Usage: python tools/make_profile_summary.py [tag]"""
import csv, glob, json, os, shutil, subprocess, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", "prof_" + tag); dst = os.path.join(R, "profiles")
shutil.copy(glob.glob(os.path.join(src, "stats", "**", "bench_kernel_stats.csv"), recursive=True)[0], os.path.join(dst, tag + "_bench_kernel_stats.csv"))
pp = glob.glob(os.path.join(src, "stats_pipe", "**", "bench_kernel_stats.csv"), recursive=True)
if pp: shutil.copy(pp[0], os.path.join(dst, tag + "_bench_pipelined_kernel_stats.csv"))      # the default command (two batches in flight)
pc = glob.glob(os.path.join(src, "stats_c2", "**", "c2_kernel_stats.csv"), recursive=True)
if pc: shutil.copy(pc[0], os.path.join(dst, tag + "_config2_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "config_rates.json")): shutil.copy(os.path.join(src, "config_rates.json"), os.path.join(dst, tag + "_config_rates.json"))
for name in ("bench_under_rocprof", "bench_pipelined_under_rocprof", "bench_line", "stream_cycles", "bench_config2", "bench_p1", "bench_p2", "bench_p4", "bench_two_pass_channel", "c_host_pipeline3"):
    if not os.path.exists(os.path.join(src, name + ".json")): continue
    line = [l for l in open(os.path.join(src, name + ".json")) if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(dst, f"{tag}_{name}.json"), "w"), indent=1)

for name, ext in (("rx2_census", "json"), ("rx2_counters", "json"), ("phase_timing_rx2", "txt"), ("rx2_census", "txt"), ("single_stream_latency", "txt")):
    f = os.path.join(src, f"{name}.{ext}")
    if os.path.exists(f): shutil.copy(f, os.path.join(dst, f"{tag}_{name}.{ext}"))

def load(name):
    fs = glob.glob(os.path.join(src, name, "**", "pmc_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in (csv.DictReader(open(fs[0])) if fs else []):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in cnt.items()}
try:
    commit = subprocess.check_output(["git", "-C", R, "rev-parse", "--short", "HEAD"]).decode().strip()
except Exception:
    commit = "?"
out = {"commit": commit,
       "command": "rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity (separate passes: MFMA/GRBM set, FETCH_SIZE, WRITE_SIZE, two SQ sets, TCC hit/miss; tools/collect_profiles.sh)",
       "note": "FETCH_SIZE/WRITE_SIZE are the raw counters x 1024 (KB units, MI355X_MICROARCH.md); the gfx950 wide-load halving correction is NOT applied here (bench.py reports raw and corrected). GRBM_GUI_ACTIVE is summed over the 8 XCDs, so cycles = GRBM/8; MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs) / (cycles x 1024 SIMDs). SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.",
       "kernels": {}, "sq_breakdown": {}}
mf, n1 = load("pmc_mfma"); fe, n2 = load("pmc_fetch"); wr, n3 = load("pmc_write"); s1, n4 = load("pmc_sq1"); s2, n5 = load("pmc_sq2"); l2, n6 = load("pmc_l2")
txt = []
for name, (agg, n) in (("pmc_mfma", (mf, n1)), ("pmc_fetch", (fe, n2)), ("pmc_write", (wr, n3)), ("pmc_sq1", (s1, n4)), ("pmc_sq2", (s2, n5)), ("pmc_l2", (l2, n6))):
    txt.append(f"# gpurun_out/prof_{tag}/{name}/pmc_counter_collection.csv")
    for k in sorted(agg, key=lambda k: -n[k]):
        txt.append(f"{k[:42]:42s} dispatches {n[k]:5d} " + " ".join(f"{c}={v / n[k]:.4g}/disp" for c, v in sorted(agg[k].items())))
open(os.path.join(dst, tag + "_pmc_per_kernel.txt"), "w").write("\n".join(txt) + "\n")
for k in mf:
    n = n1[k]; a = mf[k]
    out["kernels"][k] = {
        "dispatches": n, "gpu_cycles_per_dispatch": a.get("GRBM_GUI_ACTIVE", 0.0) / n / 8.0,
        "mfma_busy_pct": 100.0 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0) if a.get("GRBM_GUI_ACTIVE", 0.0) else 0.0,
        "mfma_mops_f32_per_dispatch": a.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / n, "mfma_mops_f64_per_dispatch": a.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / n,
        "mfma_mops_f16_per_dispatch": a.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) / n,
        "fetch_bytes_per_dispatch": 1024.0 * fe.get(k, {}).get("FETCH_SIZE", 0.0) / max(n2.get(k, 1), 1),
        "write_bytes_per_dispatch": 1024.0 * wr.get(k, {}).get("WRITE_SIZE", 0.0) / max(n3.get(k, 1), 1)}
    if k in l2 and l2[k].get("TCC_HIT_sum", 0) + l2[k].get("TCC_MISS_sum", 0) > 0:
        out["kernels"][k]["l2_hit_rate"] = l2[k]["TCC_HIT_sum"] / (l2[k]["TCC_HIT_sum"] + l2[k]["TCC_MISS_sum"])
    if k in s1 and s1[k].get("SQ_WAVE_CYCLES", 0) > 0:
        w = s1[k]["SQ_WAVE_CYCLES"]; a1 = s1[k]; a2 = s2.get(k, {})
        sh = {"parked_s_waitcnt_s_barrier": a1.get("SQ_WAIT_ANY", 0) / w, "issue_stalled": a1.get("SQ_WAIT_INST_ANY", 0) / w, "issuing": a1.get("SQ_ACTIVE_INST_ANY", 0) / w,
              "issuing_valu": a1.get("SQ_ACTIVE_INST_VALU", 0) / w, "issuing_lds": a1.get("SQ_ACTIVE_INST_LDS", 0) / w,
              "lds_bank_conflict_share_of_lds_cycles": (a2.get("SQ_LDS_BANK_CONFLICT", 0) / a2["SQ_LDS_IDX_ACTIVE"]) if a2.get("SQ_LDS_IDX_ACTIVE") else None}
        sh = {kk: (round(v, 4) if v is not None else None) for kk, v in sh.items()}
        sh["bound"] = ("latency (s_waitcnt/s_barrier)" if sh["parked_s_waitcnt_s_barrier"] > 0.4 else "issue") + (" + valu-issue" if sh["issuing_valu"] > 0.15 else "")
        out["sq_breakdown"][k] = sh
# where the receiver kernel's bytes go (per launch).  Two different things side by side, NOT subtracted from one another any more (round 3 did, and got a
# negative "remainder"): the bytes the kernel REQUESTS by source, from the call counts of the un-profiled bench line (a model), and the bytes that
# reached HBM (FETCH_SIZE / WRITE_SIZE).  counter / model < 1 on the fetch side = the rest was served by L2 / the infinity cache (a search call's
# |Dt| surface was written ~100 k cycles earlier by the same workgroup); on the write side everything listed does go out, so counter - model = spills.
try:
    bl = json.load(open(os.path.join(dst, tag + "_bench_line.json")))
    rxk = "k_rx_sync2"
    c = bl["roofline"]["per_launch_counts"]; k = out["kernels"][rxk]; B = bl["config"]["streams_per_gpu"]
    surf = 960 * 40 * 4
    state = 25.6e3            # sizeof(rd_rx_stream): scalars, filter state, rx_buf (2112 c64), two row-sum vectors
    rd = {"filtered samples in (k_bpf_fir's output; 8 B per sample consumed)": 640.0 * c["offered_frames"],
          "|Dt| surface of the previous search call (dtcache; every search call but a stream's first after (re)entering the search state)": c["search_calls"] * surf,
          "per-stream state at launch start": B * state, "decoder latents / history (zrows, hist)": c["decoded_modem_frames"] * 960 + c["decoded_modem_frames"] / 8 * 2944 * 1.0,
          "rx_buf / row sums back from HBM after each decoder stage": c["decoded_modem_frames"] / 8 * 24576}
    wr = {"features out (algorithmic)": 144.0 * 12 * c["decoded_modem_frames"], "|Dt| surface written by every search call (dtcache)": c["search_calls"] * surf,
          "per-stream state at launch end": B * state, "decoder latents, 84-float rows, history": c["decoded_modem_frames"] * (960 + 1008) + c["decoded_modem_frames"] / 8 * 2944,
          "rx_buf / row sums parked in HBM during each decoder stage": c["decoded_modem_frames"] / 8 * 24576}
    fm, wm = sum(rd.values()), sum(wr.values())
    out["rx_sync_traffic_breakdown_bytes_per_launch"] = {"kernel": rxk, "fetch_counter": k["fetch_bytes_per_dispatch"], "write_counter": k["write_bytes_per_dispatch"],
        "fetch_requested_model": rd, "fetch_requested_model_total": fm, "fetch_counter_over_model": k["fetch_bytes_per_dispatch"] / fm,
        "write_model": wr, "write_model_total": wm, "write_counter_minus_model_is_scratch_spills": k["write_bytes_per_dispatch"] - wm,
        "algorithmic_io_bytes": 784.0 * c["offered_frames"], "ratio_raw_over_io": (k["fetch_bytes_per_dispatch"] + k["write_bytes_per_dispatch"]) / (784.0 * c["offered_frames"]),
        "ratio_raw_over_io_plus_search_state": (k["fetch_bytes_per_dispatch"] + k["write_bytes_per_dispatch"]) / (784.0 * c["offered_frames"] + 2 * c["search_calls"] * surf),
        "note": "the |Dt| surface (960 x 40 float32 = 153.6 KB) a search call leaves for the next one is state the reference's algorithm defines (dsp.py keeps Dt1/Dt2 between calls); it does not fit in LDS beside the pilot search's operands and must stay float32 for the arg-max to stay bit-exact"}
    if "k_bpf_fir" in out["kernels"]:
        kb = out["kernels"]["k_bpf_fir"]; n_rx = 8000 + (bl["config"]["frames_per_stream"] // 12) * 960 + 1152 + 1152
        out["bpf_fir_traffic_bytes_per_launch"] = {"fetch_counter": kb["fetch_bytes_per_dispatch"], "write_counter": kb["write_bytes_per_dispatch"],
            "algorithmic": {"raw samples in": 8.0 * B * n_rx, "filtered samples out": 8.0 * B * n_rx}, "ratio_counter_over_algorithmic": (kb["fetch_bytes_per_dispatch"] + kb["write_bytes_per_dispatch"]) / (16.0 * B * n_rx)}
except Exception as e:
    print("traffic breakdown skipped:", e)
json.dump(out, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    if v["gpu_cycles_per_dispatch"] > 1e5: print(f"{k[:40]:40s} n={v['dispatches']:4d} cyc={v['gpu_cycles_per_dispatch']:.3g} mfma_busy={v['mfma_busy_pct']:.1f}% fetch={v['fetch_bytes_per_dispatch']/1e6:.1f}MB write={v['write_bytes_per_dispatch']/1e6:.1f}MB l2hit={v.get('l2_hit_rate')} sq={out['sq_breakdown'].get(k)}")
