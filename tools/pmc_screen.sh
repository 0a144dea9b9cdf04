#!/bin/bash
# PMC passes over the two-stage kNN (N=1M): gpurun -- 'bash tools/pmc_screen.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_s$i -- \
      python $R/tools/knn_screen_perf.py 1000000 nocheck > $R/gpurun_out/pmc_s$i.log 2>&1
  tail -1 $R/gpurun_out/pmc_s$i.log | cut -c1-200
done
cd $R
python tools/pmc_sum.py gpurun_out knn_screen_kernel | tee gpurun_out/pmc_screen.json
