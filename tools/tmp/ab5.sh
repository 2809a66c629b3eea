timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
python tools/bench_ts_gemm.py 2>&1 | tail -12
for i in 1 2; do
for sw in 0 1; do
  GLAM_TS_SW=$sw python bench.py --steps 2000 --warmup 50 --large-batch 0 --cpu-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('sw=$sw', round(d['ms_per_step']*1e3,2), {k[:24]:round(v['avg_us'],2) for k,v in d['roofline_kernels']['kernels'].items()})"
done; done
for B in 32 256 2048 4096; do for sw in 0 1; do
  GLAM_TS_SW=$sw python bench.py --batch $B --steps 1000 --warmup 50 --large-batch 0 --cpu-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print($B, 'sw=$sw', round(d['ms_per_step']*1e3,2), {k[:24]:round(v['avg_us'],2) for k,v in d['roofline_kernels']['kernels'].items() if 'gemm' in k})"
done; done
