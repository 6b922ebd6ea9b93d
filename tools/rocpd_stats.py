#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace: per-kernel count / total / avg / min / max (us)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
name_col = "kernel_name" if "kernel_name" in scols else "display_name"
q = f"select s.{name_col}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"
rows = list(db.execute(q))
tot = sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for n, c, s, mn, mx in rows:
    print(f"{n[:70]:70s} {c:7d} {s/1e3:12.1f} {s/c/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
