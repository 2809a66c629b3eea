# round-2 evidence for bench.py's roofline line: the same command under rocprofv3 --kernel-trace (B = 1024 only, and B = 16 384
# only, in SEPARATE runs so that persistent-grid kernels are not mixed), plus the FETCH_SIZE / WRITE_SIZE passes.
R=$PWD; TAG=${1:-r2a}; cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
rocprofv3 --kernel-trace -d /tmp/p0 -o k -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-seconds 0 --large-batch 0 > $R/gpurun_out/${TAG}_bench_under_rocprof_b1024.json 2>/tmp/b0.log
python3 $R/tools/rocpd_stats.py $(ls /tmp/p0/*.db /tmp/p0/*/*.db 2>/dev/null | head -1) $R/gpurun_out/${TAG}_kernel_stats_bench_b1024.txt > /dev/null
rocprofv3 --kernel-trace -d /tmp/p1 -o k -- python3 $R/bench.py --batch 16384 --steps 40 --warmup 5 --cpu-seconds 0 --large-batch 0 --prof-reps 10 > $R/gpurun_out/${TAG}_bench_under_rocprof_b16384.json 2>/tmp/b1.log
python3 $R/tools/rocpd_stats.py $(ls /tmp/p1/*.db /tmp/p1/*/*.db 2>/dev/null | head -1) $R/gpurun_out/${TAG}_kernel_stats_bench_b16384.txt > /dev/null
head -14 $R/gpurun_out/${TAG}_kernel_stats_bench_b1024.txt | cut -c1-170
head -14 $R/gpurun_out/${TAG}_kernel_stats_bench_b16384.txt | cut -c1-170
