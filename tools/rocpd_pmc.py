#!/usr/bin/env python3
"""Per-kernel average of PMC counters from a rocprofv3 rocpd sqlite db.  usage: rocpd_pmc.py results.db [out.txt]"""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
gcol = "grid_size" if "grid_size" in cols else "0"
q = f"select kernel_name, counter_name, value, dispatch_id, {gcol} from counters_collection" if "kernel_name" in cols else None
if q is None:
    print(cols); sys.exit(1)
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for name, cn, val, did, grid in db.execute(q):
    name = re.sub(r"\(.*", "", name); name = re.sub(r"^void ", "", name)
    if name.startswith("glam::"): name = f"{name} [grid={grid}]"
    acc[name][cn] += val; cnt[name].add(did)
names = sorted({c for k in acc for c in acc[k]})
lines = [f"{'kernel':60s} {'n':>5s} " + " ".join(f"{c[:22]:>22s}" for c in names)]
for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0)):
    n = len(cnt[k])
    lines.append(f"{k[:60]:60s} {n:5d} " + " ".join(f"{acc[k][c]/n:22.0f}" for c in names))
out = "\n".join(lines); print(out)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(out + "\n")
