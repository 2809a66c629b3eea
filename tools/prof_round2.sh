# Round-2 evidence (run on the GPU box from the repo root: bash tools/prof_round2.sh TAG).  Everything lands in gpurun_out/TAG_*;
# the summaries that are quoted in DESIGN.md are copied to profiles/ afterwards.
#   1. bench.py as the driver runs it                                  -> TAG_bench.json
#   2. the same command under rocprofv3 --kernel-trace, B = 1024 only   -> TAG_kernel_stats_bench_b1024.txt
#      and B = 16 384 only (separate runs: persistent-grid kernels keep one grid for both sizes)
#   3. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (gfx950 read correction in tools/rocpd_traffic.py)
#   4. full model: three presets, throughput + kernel traces
R=$PWD; TAG=${1:-r2c}; cd /tmp && export TMPDIR=/tmp
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
rocprofv3 --kernel-trace -d /tmp/p0 -o k -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-seconds 0 --large-batch 0 > $R/gpurun_out/${TAG}_bench_under_rocprof_b1024.json 2>/tmp/b0.log
python3 $R/tools/rocpd_stats.py $(db /tmp/p0) $R/gpurun_out/${TAG}_kernel_stats_bench_b1024.txt > /dev/null
rocprofv3 --kernel-trace -d /tmp/p1 -o k -- python3 $R/bench.py --batch 16384 --steps 40 --warmup 5 --cpu-seconds 0 --large-batch 0 --prof-reps 10 > $R/gpurun_out/${TAG}_bench_under_rocprof_b16384.json 2>/tmp/b1.log
python3 $R/tools/rocpd_stats.py $(db /tmp/p1) $R/gpurun_out/${TAG}_kernel_stats_bench_b16384.txt > /dev/null
for B in 1024 16384; do
  rocprofv3 --pmc FETCH_SIZE -d /tmp/f$B -o f -- python3 $R/bench.py --batch $B --steps 10 --warmup 3 --cpu-seconds 0 --large-batch 0 --prof-reps 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d /tmp/w$B -o w -- python3 $R/bench.py --batch $B --steps 10 --warmup 3 --cpu-seconds 0 --large-batch 0 --prof-reps 3 > /dev/null 2>&1
  python3 $R/tools/rocpd_traffic.py $(db /tmp/f$B) $(db /tmp/w$B) $R/gpurun_out/${TAG}_hbm_traffic_pmc_b$B.json > $R/gpurun_out/${TAG}_hbm_traffic_pmc_b$B.txt 2>&1
done
for preset in relu model_default run_default; do
  python3 $R/tools/bench_model.py --preset $preset 2>/dev/null | tail -1
  python3 $R/tools/bench_model.py --preset $preset --batch 32 2>/dev/null | tail -1
done > $R/gpurun_out/${TAG}_bench_model.log
python3 $R/tools/bench_model.py --preset run_default --block _TripletMessage 2>/dev/null | tail -1 >> $R/gpurun_out/${TAG}_bench_model.log
python3 $R/tools/bench_model.py --preset relu --out-dim 2 --storage bf16 2>/dev/null | tail -1 >> $R/gpurun_out/${TAG}_bench_model.log
python3 $R/tools/bench_model.py --preset relu --out-dim 12 --loss bcel 2>/dev/null | tail -1 >> $R/gpurun_out/${TAG}_bench_model.log
python3 $R/tools/bench_model.py --preset relu --out-dim 617 --loss bcel 2>/dev/null | tail -1 >> $R/gpurun_out/${TAG}_bench_model.log
for preset in relu model_default run_default; do
  rocprofv3 --kernel-trace -d /tmp/pm_$preset -o m -- python3 $R/tools/bench_model.py --preset $preset --steps 50 > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py $(db /tmp/pm_$preset) $R/gpurun_out/${TAG}_kernel_stats_model_$preset.txt > /dev/null
done
cut -c1-220 $R/gpurun_out/${TAG}_bench_model.log
head -12 $R/gpurun_out/${TAG}_kernel_stats_bench_b1024.txt | cut -c1-170
cat $R/gpurun_out/${TAG}_hbm_traffic_pmc_b1024.txt | head -14
