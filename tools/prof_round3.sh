# Round-3 evidence (run on the GPU box from the repo root: bash tools/prof_round3.sh TAG).  Everything lands in gpurun_out/TAG_*; the
# summaries quoted in DESIGN.md are copied to profiles/ afterwards.
#   1. bench.py as the driver runs it (default flags, and --steps 20 --warmup 5)          -> TAG_bench.json, TAG_bench_steps20.json
#   2. the same command under rocprofv3 --kernel-trace, B = 1024 and B = 16 384 in separate runs -> TAG_kernel_stats_bench_b*.txt
#   3. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (gfx950 read correction in tools/rocpd_traffic.py)
#   4. SQ counters of the step's kernels (matrix-pipe busy, issue stalls, LDS conflicts) at both sizes -> TAG_pmc_step_b*.txt
R=$PWD; TAG=${1:-r3a}; cd /tmp && export TMPDIR=/tmp
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --large-batch 0 > $R/gpurun_out/${TAG}_bench_steps20.json 2> /tmp/bench2.err || tail -5 /tmp/bench2.err
rocprofv3 --kernel-trace -d /tmp/p0 -o k -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-seconds 0 --large-batch 0 > $R/gpurun_out/${TAG}_bench_under_rocprof_b1024.json 2>/tmp/b0.log
python3 $R/tools/rocpd_stats.py $(db /tmp/p0) $R/gpurun_out/${TAG}_kernel_stats_bench_b1024.txt > /dev/null
rocprofv3 --kernel-trace -d /tmp/p1 -o k -- python3 $R/bench.py --batch 16384 --steps 40 --warmup 5 --cpu-seconds 0 --large-batch 0 --prof-reps 10 > $R/gpurun_out/${TAG}_bench_under_rocprof_b16384.json 2>/tmp/b1.log
python3 $R/tools/rocpd_stats.py $(db /tmp/p1) $R/gpurun_out/${TAG}_kernel_stats_bench_b16384.txt > /dev/null
for B in 1024 16384; do
  rocprofv3 --pmc FETCH_SIZE -d /tmp/f$B -o f -- python3 $R/bench.py --batch $B --steps 10 --warmup 3 --cpu-seconds 0 --large-batch 0 --prof-reps 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d /tmp/w$B -o w -- python3 $R/bench.py --batch $B --steps 10 --warmup 3 --cpu-seconds 0 --large-batch 0 --prof-reps 3 > /dev/null 2>&1
  python3 $R/tools/rocpd_traffic.py $(db /tmp/f$B) $(db /tmp/w$B) $R/gpurun_out/${TAG}_hbm_traffic_pmc_b$B.json > $R/gpurun_out/${TAG}_hbm_traffic_pmc_b$B.txt 2>&1
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d /tmp/s${B}_$i -o c -- python3 $R/bench.py --batch $B --steps 10 --warmup 3 --cpu-seconds 0 --large-batch 0 --prof-reps 3 > /dev/null 2>&1
    python3 $R/tools/rocpd_pmc.py $(db /tmp/s${B}_$i) /tmp/s${B}_$i.txt > /dev/null 2>&1
    grep -E "^kernel|glam::" /tmp/s${B}_$i.txt | cut -c1-330 >> $R/gpurun_out/${TAG}_pmc_step_b$B.txt
  done
done
head -12 $R/gpurun_out/${TAG}_kernel_stats_bench_b1024.txt | cut -c1-170
head -12 $R/gpurun_out/${TAG}_kernel_stats_bench_b16384.txt | cut -c1-170
cat $R/gpurun_out/${TAG}_hbm_traffic_pmc_b1024.txt | head -14
