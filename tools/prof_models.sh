#!/bin/bash
# Full-model numbers (run on the GPU box from the repo root: bash tools/prof_models.sh TAG)
R=$PWD; TAG=${1:-r5a}; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
{
for preset in relu model_default run_default; do
  python3 $R/tools/bench_model.py --preset $preset 2>/dev/null | tail -1
  python3 $R/tools/bench_model.py --preset $preset --batch 32 2>/dev/null | tail -1
done
python3 $R/tools/bench_model.py --preset run_default --block _TripletMessage 2>/dev/null | tail -1
python3 $R/tools/bench_model.py --preset relu --out-dim 2 2>/dev/null | tail -1
python3 $R/tools/bench_model.py --preset relu --out-dim 2 --batch 642 2>/dev/null | tail -1
python3 $R/tools/bench_model.py --preset relu --out-dim 2 --batch 2039 2>/dev/null | tail -1
python3 $R/tools/bench_model.py --preset relu --out-dim 12 --loss bcel 2>/dev/null | tail -1
python3 $R/tools/bench_model.py --preset relu --out-dim 617 --loss bcel 2>/dev/null | tail -1
python3 $R/tools/bench_dti.py 2>/dev/null | tail -1
} > $R/gpurun_out/${TAG}_bench_model.log
for preset in relu model_default run_default; do
  rm -rf /tmp/pm_$preset; rocprofv3 --kernel-trace -d /tmp/pm_$preset -o m -- python3 $R/tools/bench_model.py --preset $preset --steps 50 > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py $(db /tmp/pm_$preset) $R/gpurun_out/${TAG}_kernel_stats_model_$preset.txt > /dev/null
done
cut -c1-230 $R/gpurun_out/${TAG}_bench_model.log
head -25 $R/gpurun_out/${TAG}_kernel_stats_model_run_default.txt | cut -c1-150
