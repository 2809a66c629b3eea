#!/bin/bash
# The round's evidence (GPU box, repo root: bash tools/prof_round.sh TAG [stages]).  Output: gpurun_out/TAG_*; summaries quoted in DESIGN.md are copied
# to profiles/.  Stages (default "bench stats pmc"):
#   test   python -m pytest tests -m gpu -x -q                                   -> TAG_pytest_gpu.txt
#   bench  bench.py (default flags) and the driver's --steps 20 --warmup 5 form  -> TAG_bench.json, TAG_bench_steps20.json
#   stats  the same command under rocprofv3 --kernel-trace, B = 1024 / 16 384    -> TAG_kernel_stats_bench_b*.txt
#   hbm    FETCH_SIZE / WRITE_SIZE in SEPARATE --pmc passes                      -> TAG_hbm_traffic_pmc_b*.{json,txt}
#   pmc    SQ counters of the step's kernels (two passes per size)               -> TAG_pmc_step_b*.txt
#   lds    only the LDS / issue-stall pass of `pmc` (quick A/B of a layout change)
set -u
R=$PWD; TAG=${1:-r5a}; [ $# -gt 0 ] && shift; ST="${*:-bench stats pmc}"
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
prof() { local d=$1; shift; rm -rf $d; rocprofv3 "$@" || { echo "rocprofv3 pass into $d failed" >&2; exit 1; }; }
has() { case " $ST " in *" $1 "*) return 0;; esac; return 1; }
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
SMALL="--steps 10 --warmup 3 --cpu-seconds 0 --large-batch 0 --prof-reps 3"
if has test; then (cd $R && timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $R/gpurun_out/${TAG}_pytest_gpu.txt; tail -3 $R/gpurun_out/${TAG}_pytest_gpu.txt); fi
if has bench; then
  python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
  python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --large-batch 0 > $R/gpurun_out/${TAG}_bench_steps20.json 2> /tmp/bench2.err || tail -5 /tmp/bench2.err
  python3 - <<PY
import json
for f in ("$R/gpurun_out/${TAG}_bench.json", "$R/gpurun_out/${TAG}_bench_steps20.json"):
    try:
        r = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], "us/step", round(r["ms_per_step"] * 1e3, 2), "frac", r.get("roofline", {}).get("frac"))
    for k, v in r.get("roofline_kernels", {}).get("kernels", {}).items():
        print("   %-34s %7.2f us  hbm %s  mfma %s" % (k, v["avg_us"], v.get("frac_hbm_peak"), v.get("frac_mfma_f32_peak")))
    rl = r.get("roofline_large")
    if rl:
        print("  large: frac", rl.get("frac"), "sum_us", rl.get("sum_kernel_us_per_step"))
        for k, v in rl["step_kernels"].items():
            print("   %-34s %7.2f us  hbm %s  mfma %s" % (k, v["avg_us"], v.get("frac_hbm_peak"), v.get("frac_mfma_f32_peak")))
PY
fi
if has stats; then
  prof /tmp/p0 --kernel-trace -d /tmp/p0 -o k -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-seconds 0 --large-batch 0 > $R/gpurun_out/${TAG}_bench_under_rocprof_b1024.json 2>/tmp/b0.log
  python3 $R/tools/rocpd_stats.py $(db /tmp/p0) $R/gpurun_out/${TAG}_kernel_stats_bench_b1024.txt > /dev/null
  prof /tmp/p1 --kernel-trace -d /tmp/p1 -o k -- python3 $R/bench.py --batch 16384 --steps 40 --warmup 5 --cpu-seconds 0 --large-batch 0 --prof-reps 10 > $R/gpurun_out/${TAG}_bench_under_rocprof_b16384.json 2>/tmp/b1.log
  python3 $R/tools/rocpd_stats.py $(db /tmp/p1) $R/gpurun_out/${TAG}_kernel_stats_bench_b16384.txt > /dev/null
  head -12 $R/gpurun_out/${TAG}_kernel_stats_bench_b1024.txt | cut -c1-170
  head -12 $R/gpurun_out/${TAG}_kernel_stats_bench_b16384.txt | cut -c1-170
fi
for B in 1024 16384; do
  if has hbm; then
    prof /tmp/f$B --pmc FETCH_SIZE -d /tmp/f$B -o f -- python3 $R/bench.py --batch $B $SMALL > /dev/null 2>&1
    prof /tmp/w$B --pmc WRITE_SIZE -d /tmp/w$B -o w -- python3 $R/bench.py --batch $B $SMALL > /dev/null 2>&1
    python3 $R/tools/rocpd_traffic.py $(db /tmp/f$B) $(db /tmp/w$B) $R/gpurun_out/${TAG}_hbm_traffic_pmc_b$B.json > $R/gpurun_out/${TAG}_hbm_traffic_pmc_b$B.txt 2>&1
    head -14 $R/gpurun_out/${TAG}_hbm_traffic_pmc_b$B.txt
  fi
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    if has pmc || { has lds && [ $i = 2 ]; }; then
      [ $i = 1 ] && rm -f $R/gpurun_out/${TAG}_pmc_step_b$B.txt
      prof /tmp/s${B}_$i --pmc $set -d /tmp/s${B}_$i -o c -- python3 $R/bench.py --batch $B $SMALL > /dev/null 2>&1
      python3 $R/tools/rocpd_pmc.py $(db /tmp/s${B}_$i) /tmp/s${B}_$i.txt > /dev/null 2>&1
      grep -E "^kernel|glam::" /tmp/s${B}_$i.txt | cut -c1-330 >> $R/gpurun_out/${TAG}_pmc_step_b$B.txt
    fi
  done
  if has pmc || has lds; then cut -c1-60,100-330 $R/gpurun_out/${TAG}_pmc_step_b$B.txt | head -30; fi
done
