#!/bin/bash
# PMC passes over the scheduled UMAP kernels: gpurun -- 'bash tools/pmc_sched.sh "1 2 3 4 5 6 7 8"'
# (separate --pmc runs, --kernel-trace only, as the guide prescribes; summaries go to gpurun_out/pmc_sched.txt)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
SEL=${1:-"1 2 3 4 5"}
GEOM=${2:-0}
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_TA_BUSY GRBM_TC_BUSY" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_BUSY_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_CYCLE_sum" \
           "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  case " $SEL " in *" $i "*) ;; *) continue;; esac
  ITERS=8 timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_s$i -- \
      python $R/tools/umap_sched_perf.py 1000000 $GEOM 2 > $R/gpurun_out/pmc_s$i.log 2>&1
done
cd $R
for k in umap_sched_grad umap_sched_build_kernel; do echo "== $k"; for i in 1 2 3 4 5 6 7 8 9; do [ -d gpurun_out/pmc_s$i ] && python tools/pmc_sum.py gpurun_out/pmc_s$i "$k" | tr -d '\n'; echo; done; done | tee gpurun_out/pmc_sched.txt
