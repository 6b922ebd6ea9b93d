#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (sum over dispatches, mean per dispatch)."""
import csv, sys, collections
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    print(f"# {path}")
    for k in sorted(agg, key=lambda k: -len(cnt[k])):
        n = len(cnt[k])
        print(f"{k:42s} dispatches {n:5d} " + " ".join(f"{c}={v/n:.4g}/disp" for c, v in sorted(agg[k].items())))
