R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pd0 -o m -- python3 $R/tools/bench_model.py --block _NNConv --norm _PairNorm > /tmp/md.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/pd0/*.db /tmp/pd0/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1z_kernel_stats_model_nnconv_pairnorm.txt > /dev/null
tail -1 /tmp/md.log | cut -c1-300
