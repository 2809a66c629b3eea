R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/ps -o m -- python3 $R/tools/bench_model.py --readout Set2Set > /tmp/ms.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/ps/*.db /tmp/ps/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1z_kernel_stats_model_set2set.txt > /dev/null
tail -1 /tmp/ms.log | cut -c1-300
