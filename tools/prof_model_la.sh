R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pl -o m -- python3 $R/tools/bench_model.py --readout GlobalLAPool > /tmp/ml.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pl/*.db /tmp/pl/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1z_kernel_stats_model_lapool.txt > /dev/null
tail -1 /tmp/ml.log | cut -c1-300
