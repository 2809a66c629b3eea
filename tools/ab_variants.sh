for l in "" skew1 skew2; do
  if [ -n "$l" ]; then export GLAM_HIP_LIB=$PWD/glam_amd/variants/lib_$l.so; fi
  python bench.py --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', d['ms_per_step'], {k:round(v['avg_us'],2) for k,v in d['roofline_kernels']['kernels'].items() if 'dst' in k})"
done
