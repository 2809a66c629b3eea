R=$PWD; A=${1:-3}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pm$A -o m -- python3 $R/tools/bench_model.py --alpha $A > /tmp/m$A.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pm$A/*.db /tmp/pm$A/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1w_kernel_stats_model_alpha$A.txt > /dev/null
tail -1 /tmp/m$A.log | cut -c1-300
