for B in 2048 4096 8192; do
for lib in glam_amd/variants/lib_before.so glam_amd/libglam_hip.so; do
  GLAM_HIP_LIB=$PWD/$lib python bench.py --batch $B --steps 1000 --warmup 50 --large-batch 0 --cpu-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print($B, '$lib'[-12:], round(d['ms_per_step']*1e3,2), {k[:24]:round(v['avg_us'],2) for k,v in d['roofline_kernels']['kernels'].items()})"
done; done
