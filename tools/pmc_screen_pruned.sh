#!/bin/bash
# PMC passes over the cluster-pruned two-stage kNN (N=1M, headline data): gpurun -- 'bash tools/pmc_screen_pruned.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_p$i -- \
      python $R/tools/knn_screen_perf.py 1000000 nocheck > $R/gpurun_out/pmc_p$i.log 2>&1
  tail -1 $R/gpurun_out/pmc_p$i.log | cut -c1-200
done
cd $R
for i in 1 2 3; do python tools/pmc_pick.py gpurun_out/pmc_p$i "knn_screen_kernel<8, 1, 1, 3>"; done | tee gpurun_out/pmc_screen_pruned.txt
