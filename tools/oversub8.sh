#!/bin/bash
# Runs on the GPU box (gpurun): the host path of an 8-GPU job on the ONE GPU there is (VERDICT r05 #5b).  Eight real processes x three engines
# (`bench.py --gpus 8 --oversubscribe-device 0`: gloo for the two collectives, every rank on device 0) confined to 16 CPUs -- the quota seen on this pool's leases --
# contend for the host the way a node's ranks would; the single-rank run on the same box is the reference.  A host-contention proxy, NOT a scaling measurement.
# -> gpurun_out/r06_oversub8.json (copied to profiles/ by hand)
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-40}
python bench.py --steps $S --no-cpu-baseline --no-roofline --no-parity > gpurun_out/_ov_single.json 2> gpurun_out/_ov_single.err
taskset -c 0-15 python bench.py --gpus 8 --oversubscribe-device 0 --steps $S --no-cpu-baseline --no-roofline --no-parity > gpurun_out/_ov_eight.json 2> gpurun_out/_ov_eight.err
taskset -c 0-15 env RADE_SYNC=spin python bench.py --gpus 8 --oversubscribe-device 0 --steps $S --no-cpu-baseline --no-roofline --no-parity > gpurun_out/_ov_eight_spin.json 2> gpurun_out/_ov_eight_spin.err
# every shard of config 4's 2048 utterances by a single process (what rank r must reproduce bit for bit)
for r in 0 1 2 3 4 5 6 7; do python bench.py --as-shard $r/8 --steps $S --repeats 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/_ov_shard$r.json 2>/dev/null; done
python - <<'PY'
import json
def last(p):
    try: return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e: return {"error": repr(e), "stderr_tail": open(p.replace(".json", ".err")).read()[-1500:]}
a, b, c = last("gpurun_out/_ov_single.json"), last("gpurun_out/_ov_eight.json"), last("gpurun_out/_ov_eight_spin.json")
out = {"note": "8 ranks x 3 engines on ONE MI355X, confined to 16 CPUs (taskset -c 0-15): host-contention proxy for an 8-GPU node, not a scaling measurement",
       "single_rank": {k: a.get(k) for k in ("value", "ms_per_step", "host", "last_step_features_sha256_rank0", "value_repeats", "error", "stderr_tail")},
       "eight_ranks_one_device_16cpus": {k: b.get(k) for k in ("value", "ms_per_step", "host", "oversubscribed", "last_step_features_sha256_rank0", "last_step_features_sha256_per_rank", "per_rank_ms_per_step", "value_repeats", "launcher", "error", "stderr_tail")},
       "eight_ranks_one_device_16cpus_spinning": {k: c.get(k) for k in ("value", "ms_per_step", "host", "oversubscribed", "value_repeats", "error", "stderr_tail")}}
if "value" in a and "value" in b:
    out["aggregate_vs_single_rank"] = b["value"] / a["value"]
    out["rank0_bit_equal_to_single_rank"] = a["last_step_features_sha256_rank0"] == b["last_step_features_sha256_rank0"]
    out["bars"] = {"aggregate >= 0.90 of single rank": out["aggregate_vs_single_rank"] >= 0.90, "cpu cores busy (all ranks) <= 16": (b.get("oversubscribed") or {}).get("cpu_cores_busy_all_ranks", 99) <= 16.0,
                   "rx_waits_blocking > 0": b["host"]["rx_waits_blocking"] > 0, "rank 0 bit-equal": out["rank0_bit_equal_to_single_rank"]}
try:
    shards = [last(f"gpurun_out/_ov_shard{r}.json") for r in range(8)]
    out["config4_shards_single_process_sha256"] = [x.get("last_step_features_sha256_rank0") for x in shards]
    out["config4_utterances_per_rank"] = [x.get("utterances") for x in shards]
    ps = [x.get("parity_sample") or {} for x in shards]      # bench.py's parity leg on every shard: four streams each through the CPU oracle
    out["config4_oracle_parity_per_shard"] = [{k: p.get(k) for k in ("streams", "discrete_equal", "timed_equals_replay_bitwise", "feat_rms_max", "loss_delta_max", "calls_compared", "fmax_bit_equal")} for p in ps]
    out["bars"]["oracle parity on 4 streams of every shard (discrete outputs equal, features < 1e-4 RMS)"] = all(p.get("discrete_equal") and p.get("feat_rms_max", 1) < 1e-4 for p in ps)
    out["config4_all_2048_utterances_sharded_equal_single_process"] = out["config4_shards_single_process_sha256"] == b.get("last_step_features_sha256_per_rank")
    out["bars"]["every rank's shard bit-equal to a single process on that shard (config 4: 8 x 256 utterances)"] = out["config4_all_2048_utterances_sharded_equal_single_process"]
except Exception as e:
    out["config4_shard_check_error"] = repr(e)
if "value" in a and "value" in c: out["aggregate_vs_single_rank_spinning"] = c["value"] / a["value"]
json.dump(out, open("gpurun_out/r06_oversub8.json", "w"), indent=1)
print(json.dumps({k: out.get(k) for k in ("aggregate_vs_single_rank", "aggregate_vs_single_rank_spinning", "rank0_bit_equal_to_single_rank", "config4_all_2048_utterances_sharded_equal_single_process", "config4_shard_check_error", "bars")}))
print(json.dumps(out["eight_ranks_one_device_16cpus"])[:1500])
PY
