cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for case in ${CASES:-identity scaled}; do
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set -d /tmp/pmc_${case}_$tag -- python $R/scripts/pmc_case.py $case > /dev/null 2>&1
  echo "== $case : $set"
  python $R/scripts/pmc_by_kernel.py /tmp/pmc_${case}_$tag "k_fused"
done; done
