# SQ counters of the fused forward kernels (run on the GPU box from the repo root: bash tools/pmc_fwd.sh TAG B)
R=$PWD; TAG=${1:-r3}; B=${2:-16384}; cd /tmp && export TMPDIR=/tmp
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
export GLAM_PIPE_FUSED=1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pmc_${TAG}_$i -o c -- python3 $R/tools/run_fwd_only.py $B 5 > /tmp/pmc_$i.log 2>&1 || tail -3 /tmp/pmc_$i.log
  python3 $R/tools/rocpd_pmc.py $(db /tmp/pmc_${TAG}_$i) /tmp/pmc_${TAG}_$i.txt > /dev/null 2>&1 || echo "pmc pass $i failed"
  grep -E "kernel|k_triplet_fwd|k_ts_gemm" /tmp/pmc_${TAG}_$i.txt >> $R/gpurun_out/${TAG}_pmc_fwd_b$B.txt
done
cat $R/gpurun_out/${TAG}_pmc_fwd_b$B.txt
