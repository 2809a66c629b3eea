#!/usr/bin/env python3
"""Developer aid: per-kernel VGPR / scratch / occupancy table from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  usage: resusage.py file.hip"""
import re, subprocess, sys
src = sys.argv[1]
p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
blocks = re.split(r"Function Name: ", p.stderr)[1:]
names = [b.split("\n")[0].split(" ")[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
rows = set()
for b, n in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    n = re.sub(r"void glam::", "", n).split("(")[0]
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    rows.add(f"{n:58s} vgpr={g('VGPRs'):4d} agpr={g('AGPRs'):3d} sgpr={g('SGPRs'):3d} scratch={scratch:5d} occ={occ} lds={lds}")
print("\n".join(sorted(rows)))
