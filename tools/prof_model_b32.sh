# Kernel trace of the full-model training step at the reference's batch size of 32 (bash tools/prof_model_b32.sh [TAG] [PRESET]).
R=$PWD; TAG=${1:-r2i}; PRESET=${2:-model_default}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pb -o m -- python3 $R/tools/bench_model.py --batch 32 --steps 200 --preset $PRESET > /tmp/mb.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pb/*.db /tmp/pb/*/*.db 2>/dev/null | head -1) $R/gpurun_out/${TAG}_kernel_stats_model_b32_$PRESET.txt > /dev/null
tail -1 /tmp/mb.log | cut -c1-300
