# full-model numbers of round 2: parity preset, the reference's two default configurations in training mode, + kernel traces
R=$PWD; TAG=${1:-r2b}; cd /tmp && export TMPDIR=/tmp
for preset in relu model_default run_default; do
  python3 $R/tools/bench_model.py --preset $preset 2>/dev/null | tail -1
  python3 $R/tools/bench_model.py --preset $preset --batch 32 2>/dev/null | tail -1
done > $R/gpurun_out/${TAG}_bench_model.log
python3 $R/tools/bench_model.py --preset run_default --block _TripletMessage 2>/dev/null | tail -1 >> $R/gpurun_out/${TAG}_bench_model.log
for preset in relu model_default run_default; do
  rocprofv3 --kernel-trace -d /tmp/pm_$preset -o m -- python3 $R/tools/bench_model.py --preset $preset --steps 50 > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py $(ls /tmp/pm_$preset/*.db /tmp/pm_$preset/*/*.db 2>/dev/null | head -1) $R/gpurun_out/${TAG}_kernel_stats_model_$preset.txt > /dev/null
done
cat $R/gpurun_out/${TAG}_bench_model.log
head -30 $R/gpurun_out/${TAG}_kernel_stats_model_model_default.txt | cut -c1-160
