# Developer aid: the headline step under runtime environment knobs of the HIP runtime (names from `strings libamdhip64.so`).
# ROC_SYSTEM_SCOPE_SIGNAL=0 is left out: the process hangs.  Result (round 2): nothing moves the 91.7 us step by more than its noise;
# AMD_OPT_FLUSH=0 (system-scope fences) costs 14 us, i.e. the default already uses device-scope fences between the kernels.
R=$PWD
run() { echo -n "$1 : "; env $1 timeout 90 python3 $R/bench.py --cpu-seconds 0 --large-batch 0 --steps 4000 --warmup 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,2),'us')" || echo failed/timeout; }
run X=1
for v in DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=64 DEBUG_HIP_GRAPH_BATCH_SIZE=256 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 GPU_FLUSH_ON_EXECUTION=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1 DEBUG_HIP_KERNARG_COPY_OPT=0 DEBUG_HIP_KERNARG_COPY_OPT=1 ROC_SKIP_KERNEL_ARG_COPY=1 HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1 ROC_ACTIVE_WAIT_TIMEOUT=100 GPU_MAX_HW_QUEUES=1 GPU_MAX_HW_QUEUES=2 AMD_DIRECT_DISPATCH=0 DEBUG_HIP_DYNAMIC_QUEUES=0; do run $v; done
run X=1
