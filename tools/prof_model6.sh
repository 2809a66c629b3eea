R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pm -o m -- python3 $R/tools/bench_model.py --alpha 6 > /tmp/m.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pm/*.db /tmp/pm/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1w_kernel_stats_model_alpha6.txt > /dev/null
tail -1 /tmp/m.log | cut -c1-300
