R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pb -o m -- python3 $R/tools/bench_model.py --batch 32 --steps 200 > /tmp/mb.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pb/*.db /tmp/pb/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1y_kernel_stats_model_b32.txt > /dev/null
tail -1 /tmp/mb.log | cut -c1-300
