# k_param_grads<SETS> (the operand-set instantiation) role by role under rocprofv3, in the full-model step (GPU box): see tools/README.md
R=/root/repo; cd /tmp; export TMPDIR=/tmp
for r in ${ROLES:-0 1 2 3}; do
  export GLAM_HIP_LIB=$R/tools/tmp/variants/lib_pg_only$r.so
  rm -rf /tmp/pr_$r
  rocprofv3 --kernel-trace -d /tmp/pr_$r -o m -- python3 $R/tools/bench_model.py --preset relu --steps 30 > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py $(ls /tmp/pr_$r/*.db /tmp/pr_$r/*/*.db 2>/dev/null | head -1) /tmp/st_$r.txt > /dev/null
  echo "role $r:"; grep "k_param_grads" /tmp/st_$r.txt | cut -c1-60,100-160
done
