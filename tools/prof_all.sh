R=$PWD; cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 300 --warmup 30 > $R/gpurun_out/r1x_bench.json 2> /tmp/bench.err
rocprofv3 --kernel-trace -d /tmp/p0 -o k -- python3 $R/bench.py --steps 100 --warmup 10 --cpu-seconds 0.2 > /tmp/b0.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/p0/*.db /tmp/p0/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1x_kernel_stats_bench.txt > /dev/null
rocprofv3 --pmc FETCH_SIZE -d /tmp/p1 -o f -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0.2 > /tmp/b1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p2 -o w -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0.2 > /tmp/b2.log 2>&1
python3 $R/tools/rocpd_traffic.py $(ls /tmp/p1/*.db /tmp/p1/*/*.db 2>/dev/null | head -1) $(ls /tmp/p2/*.db /tmp/p2/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1x_hbm_traffic_pmc.json > $R/gpurun_out/r1x_hbm_traffic_pmc.txt 2>&1
rocprofv3 --kernel-trace -d /tmp/p3 -o m -- python3 $R/tools/bench_model.py > $R/gpurun_out/r1x_bench_model.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls /tmp/p3/*.db /tmp/p3/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1x_kernel_stats_full_model.txt > /dev/null
tail -3 $R/gpurun_out/r1x_bench_model.log; cat $R/gpurun_out/r1x_bench.json | cut -c1-400
