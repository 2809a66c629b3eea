# A/B of library variants on the headline step: tools/ab_variants.sh <variant.so> ...  ("" = the default build)
for l in "" "$@"; do
  if [ -n "$l" ]; then export GLAM_HIP_LIB=$PWD/tools/tmp/variants/$l; fi
  python bench.py --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', round(d['ms_per_step']*1e3,2), {k:round(v['avg_us'],2) for k,v in d['roofline_kernels']['kernels'].items() if 'dst' in k})"
done
