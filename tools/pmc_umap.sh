#!/bin/bash
# PMC passes over the UMAP gradient kernels: gpurun -- 'bash tools/pmc_umap.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
SEL=${1:-"1 2 3"}
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum"; do
  i=$((i+1))
  case " $SEL " in *" $i "*) ;; *) continue;; esac
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_u$i -- \
      python $R/tools/umap_perf.py 1000000 > $R/gpurun_out/pmc_u$i.log 2>&1
done
cd $R
for k in umap_neg_dense umap_neg_slice "umap_grad_kernel<2, 16, 4, true>"; do echo "== $k"; for i in 1 2 3; do python tools/pmc_sum.py gpurun_out/pmc_u$i "$k" | tr -d '\n'; echo; done; done
