R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pd -o d -- python3 $R/tools/bench_dti.py 32 > /tmp/d.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pd/*.db /tmp/pd/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1u_kernel_stats_dti.txt > /dev/null
tail -2 /tmp/d.log | cut -c100-300
