#!/bin/bash
# PMC passes over the threshold-scan kernel (N = 1M, one term): gpurun -- 'bash tools/pmc_flat.sh [terms]'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-1}
O=$R/gpurun_out/pmc_flat_${T}; mkdir -p $O
cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -- python $R/tools/knn_flat_pmc.py 1000000 $T > $O/p$i.log 2>&1
  tail -2 $O/p$i.log | cut -c1-200
done
cd $R
python tools/pmc_sum.py $O knn_flat_scan_kernel | tee $O/sum.json
python - <<PY
import csv, glob
for f in glob.glob("$O/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "knn_flat_scan_kernel" in r["Kernel_Name"]:
            print("duration_ns", int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "vgpr", r.get("VGPR_Count"), "lds", r.get("LDS_Block_Size"))
PY
