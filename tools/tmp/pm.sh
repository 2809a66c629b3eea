R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm; rocprofv3 --kernel-trace -d /tmp/pm -o m -- python3 $R/tools/bench_model.py --preset relu --steps 50 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pm/*.db /tmp/pm/*/*.db 2>/dev/null | head -1) $R/gpurun_out/tmp_model_relu.txt > /dev/null
head -40 $R/gpurun_out/tmp_model_relu.txt | cut -c1-150
