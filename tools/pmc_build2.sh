#!/bin/bash
# PMC passes over umap_sched_build2_kernel (windows of 32 iterations at N = 1M, production numbering):
#   gpurun --timeout 900 -- 'bash tools/pmc_build2.sh [stage]'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
STAGE=${1:-0}
mkdir -p $R/gpurun_out
cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "TCC_REQ_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "WRITE_SIZE" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_b2_$i -- \
      python $R/tools/sched_build2_perf.py 1000000 --one $STAGE > $R/gpurun_out/pmc_b2_$i.log 2>&1
done
cd $R
for i in 1 2 3 4 5 6; do [ -d gpurun_out/pmc_b2_$i ] && python tools/pmc_sum.py gpurun_out/pmc_b2_$i umap_sched_build2_kernel | tr -d '\n'; echo; done | tee gpurun_out/pmc_build2.txt
