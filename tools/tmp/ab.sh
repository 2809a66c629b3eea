timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
for lib in glam_amd/variants/lib_before.so glam_amd/libglam_hip.so; do
  GLAM_HIP_LIB=$PWD/$lib python bench.py --steps 2000 --warmup 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$lib', round(d['ms_per_step']*1e3,2), {k[:24]:round(v['avg_us'],2) for k,v in d['roofline_kernels']['kernels'].items()}, 'large', {k[:24]:round(v['avg_us'],1) for k,v in d['roofline_large']['step_kernels'].items()})"
done; done
