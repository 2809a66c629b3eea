R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pm -o m -- python3 $R/tools/bench_model.py --preset model_default --steps 50 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pm/*.db /tmp/pm/*/*.db 2>/dev/null | head -1) $R/gpurun_out/tmp_model_default.txt > /dev/null
head -45 $R/gpurun_out/tmp_model_default.txt | cut -c1-175
