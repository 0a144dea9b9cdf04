#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_e$i -- python $R/tools/ell_probe.py > $R/gpurun_out/pmc_e$i.log 2>&1
done
cd $R
grep "^{" gpurun_out/pmc_e1.log
for i in 1 2 3; do python tools/pmc_sum.py gpurun_out/pmc_e$i umap_sched_build_ell_kernel | tr -d '\n'; echo; done | tee gpurun_out/pmc_ell.txt
