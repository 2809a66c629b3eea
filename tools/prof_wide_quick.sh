R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pw -o w -- python3 $R/tools/bench_layer_widths.py 6 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pw/*.db /tmp/pw/*/*.db 2>/dev/null | head -1) $R/gpurun_out/tmp_wide.txt > /dev/null
head -24 $R/gpurun_out/tmp_wide.txt | cut -c1-170
