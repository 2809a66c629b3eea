R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pw -o w -- python3 $R/tools/bench_layer_widths.py 6 > /tmp/w.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pw/*.db /tmp/pw/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1v_kernel_stats_wide.txt > /dev/null
tail -1 /tmp/w.log
