# issue / wait counters of the f32x3 kernels on one layer shape:  bash scripts/x3_pmc.sh <tag> 1x1|3x3 Co Ci H W [d]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/r6/pmc_$TAG
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_WAVES_EQ_64 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pmc_x3_$i -- python $R/scripts/x3_one_layer.py "$@" > /dev/null 2> $O/err_$i.txt
  python $R/scripts/pmc_by_kernel.py /tmp/pmc_x3_$i "k_" > $O/set$i.txt
done
grep -h "k_conv\|k_wgrad" $O/set*.txt
