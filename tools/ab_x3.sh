for x in 0 1 0 1; do GLAM_X3=$x python3 bench.py --batch 16384 --steps 40 --warmup 5 --cpu-seconds 0 --large-batch 0 --prof-reps 10 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['roofline_kernels']['kernels']; print('X3=$x B=16384 step', round(r['ms_per_step']*1e3,1), {n.split('<')[0][-14:]: round(v['avg_us'],1) for n,v in k.items()})"; done
for x in 0 1 0 1; do GLAM_X3=$x python3 bench.py --steps 4000 --cpu-seconds 0 --large-batch 0 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['roofline_kernels']['kernels']; print('X3=$x B=1024 step', round(r['ms_per_step']*1e3,2), {n.split('<')[0][-14:]: round(v['avg_us'],2) for n,v in k.items()})"; done
