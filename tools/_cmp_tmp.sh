for w in 16 12 16 12; do
  for B in 16384; do
    export GLAM_TS_RB_WAVES=$w
    python bench.py --batch $B --large-batch 0 --cpu-seconds 0 --steps 300 --warmup 20 2>/dev/null | tail -1 > gpurun_out/x.json
    python3 -c "
import json,sys
d=json.load(open('gpurun_out/x.json')); rk=d['roofline_kernels']['kernels']; print('$w', $B, round(d['ms_per_step']*1000,1), [round(v['avg_us'],1) for k,v in rk.items() if 'ts_gemm' in k])"
  done
done
unset GLAM_TS_RB_WAVES
python bench.py --large-batch 0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/x.json
python3 -c "
import json,sys
d=json.load(open('gpurun_out/x.json')); rk=d['roofline_kernels']['kernels']; print(1024, round(d['ms_per_step']*1000,1), [(k[:20],round(v['avg_us'],1)) for k,v in rk.items()])"
timeout 900 python -m pytest tests -m gpu -q -x --timeout=300 -k "gemm or gru or linear or golden or layer or bf16" 2>&1 | tail -2
