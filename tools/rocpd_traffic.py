#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), grouped by kernel and
grid size.  Applies the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE
reports exactly half of the bytes of a wide (16 B/lane) coalesced read stream, so reads are doubled;
both counters are in KiB.  usage: rocpd_traffic.py fetch.db write.db [out.json]"""
import json, re, sqlite3, sys
from collections import defaultdict

def load(path, counter):
    db = sqlite3.connect(path)
    acc = defaultdict(lambda: [0.0, 0])
    for name, grid, val in db.execute("select kernel_name, grid_size, value from counters_collection where counter_name=?", (counter,)):
        name = re.sub(r"\(.*", "", name); name = re.sub(r"^void ", "", name)
        a = acc[(name, grid)]; a[0] += val; a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}

fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
rows = {}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    if not k[0].startswith("glam::"):
        continue
    rd = fetch[k] * 1024 * 2           # KiB -> B, x2 gfx950 wide-stream correction
    wr = write.get(k, 0.0) * 1024
    rows[f"{k[0]} grid={k[1]}"] = {"launches": nf[k], "fetch_size_kib_raw": fetch[k], "write_size_kib_raw": write.get(k, 0.0),
                                   "read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes": rd + wr}
    print(f"{k[0][:60]:60s} grid={k[1]:8d} n={nf[k]:4d} read={rd/1e6:9.2f} MB (x2 corrected)  write={wr/1e6:9.2f} MB  total={(rd+wr)/1e6:9.2f} MB")
if len(sys.argv) > 3:
    json.dump(rows, open(sys.argv[3], "w"), indent=1)
