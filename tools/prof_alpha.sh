#!/bin/bash
# Full-model step across the hidden widths of the search space (src_1gp/glam.py:60) — run on the GPU box: bash tools/prof_alpha.sh TAG
R=$PWD; TAG=${1:-r5a}; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
{
for a in 1 2 3 4 6; do
  python3 $R/tools/bench_model.py --preset relu --alpha $a 2>/dev/null | tail -1
done
} > $R/gpurun_out/${TAG}_bench_alpha.log
for a in 6 2; do
  rm -rf /tmp/pa_$a; rocprofv3 --kernel-trace -d /tmp/pa_$a -o m -- python3 $R/tools/bench_model.py --preset relu --alpha $a --steps 50 > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py $(db /tmp/pa_$a) $R/gpurun_out/${TAG}_kernel_stats_alpha$a.txt > /dev/null
done
cut -c1-230 $R/gpurun_out/${TAG}_bench_alpha.log
head -30 $R/gpurun_out/${TAG}_kernel_stats_alpha6.txt | cut -c1-150
head -24 $R/gpurun_out/${TAG}_kernel_stats_alpha2.txt | cut -c1-150
