#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/knn_ablate.py 400000 > gpurun_out/ablate.log 2>&1; cat gpurun_out/ablate.log | grep lib
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$name -- python $GRAFT_REPO_ROOT/tools/knn_perf.py 400000 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log 2>&1
done
cd $GRAFT_REPO_ROOT; find gpurun_out -name "*counter_collection.csv" | head
