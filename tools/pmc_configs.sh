#!/bin/bash
# rocprofv3 kernel stats + PMC passes for BASELINE configs C3 (ne_grad_kernel) and C5 (pair_scan_kernel<SeaStats>):
#   gpurun --timeout 900 -- 'bash tools/pmc_configs.sh'
# (separate --pmc runs with --kernel-trace only, as the guide prescribes; summaries in gpurun_out/pmc_configs.txt)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for c in c3 c5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$c -- python $R/tools/config_roofline.py $c > $R/gpurun_out/roof_$c.log 2>&1
done
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  for c in c3 c5; do
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${c}_$i -- python $R/tools/config_roofline.py $c > $R/gpurun_out/pmc_${c}_$i.log 2>&1
  done
done
cd $R
for c in c3 c5; do tail -1 gpurun_out/roof_$c.log | cut -c1-1200; done
{ echo "== c3 ne_grad_kernel"; for i in 1 2 3 4 5; do python tools/pmc_sum.py gpurun_out/pmc_c3_$i ne_grad_kernel | tr -d '\n'; echo; done
  echo "== c5 pair_scan_kernel"; for i in 1 2 3 4 5; do python tools/pmc_sum.py gpurun_out/pmc_c5_$i pair_scan_kernel | tr -d '\n'; echo; done; } | tee gpurun_out/pmc_configs.txt
for c in c3 c5; do f=$(ls -t gpurun_out/prof_$c/*/*kernel_stats.csv | head -1); echo "== $c $f"; head -8 "$f" | cut -c1-200; done
