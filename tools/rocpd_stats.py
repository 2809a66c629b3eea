#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table:
calls, total/avg/min/max duration (us), share of GPU kernel time.  usage: rocpd_stats.py results.db [out.txt]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
grid_col = "grid_size" if "grid_size" in cols else ("grid_x" if "grid_x" in cols else None)
rows = db.execute(f"select {name_col}, start, end, {grid_col or 0} from kernels").fetchall()
agg = {}
for name, s, e, grid in rows:
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("glam::") and grid_col:
        name = f"{name} [grid={grid}]"        # same kernel at different problem sizes is reported separately
    a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
    d = (e - s) / 1e3
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
lines = [f"{'kernel':100s} {'calls':>7s} {'total_us':>12s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"{name[:100]:100s} {a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:9.2f} {a[2]:9.2f} {a[3]:9.2f} {100*a[1]/tot:6.2f}")
lines.append(f"{'TOTAL':100s} {sum(a[0] for a in agg.values()):7d} {tot:12.1f}")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
