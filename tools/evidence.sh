#!/bin/bash
# The round's whole evidence set in one call (GPU box, repo root: bash tools/evidence.sh TAG > gpurun_out/evidence.log 2>&1).  Needs the timeline
# variant of the library (tools/build_prof_variant.sh tl, built in the container: it travels with the snapshot).  Everything lands in
# gpurun_out/TAG_*; what DESIGN.md quotes is copied to profiles/.
TAG=${1:-r6}
R=$PWD
bash tools/prof_round.sh $TAG test bench stats hbm pmc > gpurun_out/${TAG}_prof_round.log 2>&1
rm -f gpurun_out/${TAG}_bench_model.log
bash tools/prof_models.sh $TAG > gpurun_out/${TAG}_prof_models.log 2>&1
bash tools/prof_alpha.sh $TAG > gpurun_out/${TAG}_prof_alpha.log 2>&1
cd $R
python3 tools/bench_model.py --preset relu --batch 32 --steps 300 2>/dev/null | tail -1 >> gpurun_out/${TAG}_bench_model.log
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pm_b32 && rocprofv3 --kernel-trace -d /tmp/pm_b32 -o m -- python3 $R/tools/bench_model.py --preset relu --batch 32 --steps 200 > /dev/null 2>&1; python3 $R/tools/rocpd_stats.py $(ls /tmp/pm_b32/*.db /tmp/pm_b32/*/*.db 2>/dev/null | head -1) $R/gpurun_out/${TAG}_kernel_stats_model_b32_relu.txt > /dev/null )
python3 tools/bench_infer.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_bench_infer.txt
for b in 32 1024; do python3 tools/bench_graphed_trainer.py $b 2>&1 | grep -v amdgpu.ids; done > gpurun_out/${TAG}_bench_graphed_trainer.txt
for b in 1024 16384; do GLAM_HIP_LIB=$R/tools/tmp/variants/lib_tlprof.so python3 tools/ws_timeline.py $b 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_ws_timeline_b$b.txt; done
for m in GATES NODE; do echo "=== $m=1 (tools/gru_timeline.py 26000, timeline variant)"; env $m=1 GLAM_HIP_LIB=$R/tools/tmp/variants/lib_tlprof.so python3 tools/gru_timeline.py 26000 2>&1 | grep -v amdgpu.ids; done > gpurun_out/${TAG}_gru_timeline.txt
python3 tools/kernel_sequence.py 32 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_kernel_sequence_b32.txt
python3 tools/kernel_sequence.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_kernel_sequence_b1024.txt
python3 tools/kernel_sequence.py 32 model_default 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_kernel_sequence_b32_model_default.txt
python3 tools/kernel_sequence.py 1024 model_default 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_kernel_sequence_b1024_model_default.txt
# A/B of this round's switches on the model step, one box
for p in relu model_default; do for b in 32 1024; do
  for k in GLAM_X3=1 GLAM_GRU_GATES=0 GLAM_NODE_IN_GRU=0 GLAM_RRELU_IN_GEMM=0 GLAM_HEAD_ACT=0; do
    echo "$p B=$b $k: $(env $k python3 tools/bench_model.py --batch $b --steps 300 --preset $p 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*')"
  done; done; done > gpurun_out/${TAG}_ab_switches.txt
python3 tools/fork_price.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_fork_price.txt
bash tools/robust.sh > gpurun_out/${TAG}_robust.txt 2>&1
tail -40 gpurun_out/${TAG}_robust.txt
