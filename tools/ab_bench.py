#!/usr/bin/env python3
"""tools/ab_bench.py -- the A/B harness that decides (developer aid, runs on the GPU box).

Round 4's keep / reject calls of 2-3 % were taken on 2-4 rounds per variant while one library moved 69.2 -> 72.9 M frames/s between rounds on one box
(VERDICT r04, weak #5).  This harness interleaves the variants (A B C A B C ...: every round sees every variant once, in an order that rotates, so a
drift of the box's clocks hits all of them alike), runs >= 10 rounds, and reports per variant the median and the MAD (median absolute deviation) and, against
the FIRST variant (the baseline), the PAIRED per-round difference: its median, its MAD and a verdict --

    decision rule:   |median of the paired differences| >= 2 x MAD of the paired differences  (and >= 2 x MAD / sqrt(rounds) is NOT accepted: the MAD of
                     the differences themselves is the yardstick, as the verdict asked)  ->  "faster" / "slower";  otherwise "no decision".

A variant is a library (`name=path/to/lib.so`, loaded through $RADE_LIBRADEHIP), optionally with environment switches (`name=path.so,VAR=value,...`;
`name=,VAR=value` = the default library with a switch) and a command prefix (`name=,VAR=value,@taskset -c 0,1`: everything after the `@` is put in front of the
command -- note that the prefix may contain commas, so it must come last).  What is measured is chosen by --metric:
    bench    bench.py --no-cpu-baseline --no-roofline --no-parity --steps S  ->  M frames/s (higher is better)          [default]
    rx512    tools/rx_only.py 8 2 512 -> ms of ONE receiver launch of 512 streams (lower is better)
    cycles   tools/stream_cycles.py -> mean per-stream cycles of the receiver launch (lower is better)
    class:X  bench.py's roofline leg, per_class_ms_per_step[X] (e.g. class:gemm) -> ms (lower is better)

usage: python tools/ab_bench.py --rounds 10 base=ab/base.so new=radae_amd/libradehip.so [--metric bench] [--steps 60] [--out profiles/r05_ab_x.txt] [-- extra bench args]
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

R = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mad(xs):
    m = statistics.median(xs)
    return statistics.median([abs(x - m) for x in xs])


def run_one(metric, lib, envs, steps, extra, prefix=()):
    env = dict(os.environ)
    if lib:
        env["RADE_LIBRADEHIP"] = lib if os.path.isabs(lib) else os.path.join(R, lib)
    env.update(envs)
    if metric == "bench":
        cmd = list(prefix) + [sys.executable, os.path.join(R, "bench.py"), "--no-cpu-baseline", "--no-roofline", "--no-parity", "--steps", str(steps)] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        run_one.last_host = d.get("host")
        return d["value"] / 1e6
    if metric.startswith("class:"):
        cmd = [sys.executable, os.path.join(R, "bench.py"), "--no-cpu-baseline", "--no-parity", "--steps", str(steps)] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        return float(d["roofline"]["per_class_ms_per_step"][metric.split(":", 1)[1]])
    if metric == "rx512":
        out = subprocess.run([sys.executable, os.path.join(R, "tools", "rx_only.py"), "8", "2", "512"], env=env, capture_output=True, text=True).stdout
        w = out.strip().splitlines()[-1].split()
        return float(w[w.index("ms/launch") + 1])
    if metric == "cycles":
        out = subprocess.run([sys.executable, os.path.join(R, "tools", "stream_cycles.py")], env=env, capture_output=True, text=True).stdout
        return float(json.loads(out)["seeds"]["1"]["cycles_mean"])
    raise SystemExit(f"unknown metric {metric}")


def decide(base, var, higher_is_better):
    """paired per-round differences var - base (in % of the baseline's median): (median, MAD, verdict)"""
    b0 = statistics.median(base)
    d = [100.0 * (v - b) / b0 for v, b in zip(var, base)]
    med, sp = statistics.median(d), mad(d)
    if abs(med) < 2.0 * sp or sp == 0.0 and med == 0.0:
        verdict = "no decision"
    else:
        verdict = "faster" if (med > 0) == higher_is_better else "slower"
    return med, sp, verdict, d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--metric", default="bench")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--out", default=None)
    ap.add_argument("variants", nargs="+")
    args, extra = ap.parse_known_args()
    if extra and extra[0] == "--":
        extra = extra[1:]
    variants = []
    for v in args.variants:
        name, _, rest = v.partition("=")
        rest, _, pre = rest.partition("@")
        parts = rest.rstrip(",").split(",") if rest else [""]
        lib = parts[0]
        envs = dict(p.split("=", 1) for p in parts[1:] if p)
        variants.append((name, lib, envs, tuple(pre.split()) if pre else ()))
    higher = args.metric == "bench"
    unit = {"bench": "M frames/s", "rx512": "ms", "cycles": "cycles"}.get(args.metric, "ms")
    res = {name: [] for name, _, _, _ in variants}; hosts = {name: [] for name, _, _, _ in variants}
    lines = []

    def emit(s):
        print(s, flush=True); lines.append(s)

    emit(f"# tools/ab_bench.py  metric={args.metric} ({unit}, {'higher' if higher else 'lower'} is better)  rounds={args.rounds}  steps={args.steps}  extra={extra}")
    emit("# variants: " + "; ".join(f"{n} = {l or '(default library)'} {e or ''} {' '.join(pf)}" for n, l, e, pf in variants))
    t0 = time.time()
    for r in range(args.rounds):
        order = variants[r % len(variants):] + variants[:r % len(variants)]      # rotate the order: no variant always runs first (coldest) or last
        for name, lib, envs, pre in order:
            try:
                run_one.last_host = None
                x = run_one(args.metric, lib, envs, args.steps, extra, pre)
                if run_one.last_host: hosts[name].append(run_one.last_host)
            except Exception as e:          # a crashed run is recorded, not silently dropped
                emit(f"round {r + 1} {name}: FAILED {type(e).__name__} {e}")
                x = float("nan")
            res[name].append(x)
        emit(f"round {r + 1:2d}  " + "  ".join(f"{n} {res[n][-1]:.3f}" for n, _, _, _ in variants))
    emit(f"# {time.time() - t0:.0f} s")
    base_name = variants[0][0]
    ok = [i for i in range(args.rounds) if all(res[n][i] == res[n][i] for n in res)]
    emit(f"# rounds used: {len(ok)} of {args.rounds}")
    summary = {}
    for name, _, _, _ in variants:
        xs = [res[name][i] for i in ok]
        summary[name] = {"median": statistics.median(xs), "mad": mad(xs), "min": min(xs), "max": max(xs)}
        emit(f"{name:>14s}: median {summary[name]['median']:.3f} {unit}  MAD {summary[name]['mad']:.3f} ({100 * summary[name]['mad'] / summary[name]['median']:.2f} %)  range {min(xs):.3f} .. {max(xs):.3f}")
        if hosts[name]:
            hh = hosts[name]
            summary[name]["host"] = {"cpu_cores_busy_median": statistics.median(h["cpu_cores_busy"] for h in hh), "logical_cpus": hh[0]["logical_cpus"],
                                     "rx_waits_blocking": hh[0]["rx_waits_blocking"], "rx_waits_spinning": hh[0]["rx_waits_spinning"]}
            emit(f"{'':>14s}  host: {summary[name]['host']}")
    for name, _, _, _ in variants[1:]:
        med, sp, verdict, _ = decide([res[base_name][i] for i in ok], [res[name][i] for i in ok], higher)
        summary[name].update({"paired_diff_pct_median": med, "paired_diff_pct_mad": sp, "verdict": verdict})
        emit(f"{name:>14s} vs {base_name}: paired difference median {med:+.2f} %  MAD {sp:.2f} %  -> {verdict.upper()}  (rule: |median| >= 2 x MAD)")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")
        with open(os.path.splitext(args.out)[0] + ".json", "w") as f:
            json.dump({"metric": args.metric, "unit": unit, "rounds": args.rounds, "steps": args.steps, "raw": res, "summary": summary}, f, indent=1)


if __name__ == "__main__":
    main()
