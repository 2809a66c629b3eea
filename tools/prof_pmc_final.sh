# SQ issue / busy counters of every kernel of the headline step (separate --pmc passes, no trace domains)
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -d /tmp/q1 -o a -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --large-batch 0 > /tmp/q1.log 2>&1
python3 $R/tools/rocpd_pmc.py $(ls /tmp/q1/*.db /tmp/q1/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1x_pmc_sq_insts.txt > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES -d /tmp/q2 -o b -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --large-batch 0 > /tmp/q2.log 2>&1
python3 $R/tools/rocpd_pmc.py $(ls /tmp/q2/*.db /tmp/q2/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r1x_pmc_sq_mfma_lds.txt > /dev/null
tail -2 /tmp/q2.log | cut -c1-200
