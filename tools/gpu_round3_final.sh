#!/bin/bash
# Round-3 closing measurement pass on one MI355X box:  gpurun --timeout 2400 -- 'bash tools/gpu_round3_final.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/tests.log; tail -3 $O/tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 700 python bench.py --steps 5 --warmup 1 > $O/bench.log 2>&1; grep "^{" $O/bench.log | cut -c1-300
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-knn-variants > $O/bench_prof.log 2>&1
for c in c3 c5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python $R/tools/config_roofline.py $c > $O/roof_$c.log 2>&1
done
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_c5_$i -- python $R/tools/config_roofline.py c5 > $O/pmc_c5_$i.log 2>&1
done
cd $R
for c in c3 c5; do grep "^{" $O/roof_$c.log | cut -c1-1500; done
{ echo "== c5 pair_scan_kernel"; for i in 1 2 3 4 5; do python tools/pmc_sum.py $O/pmc_c5_$i pair_scan_kernel | tr -d '\n'; echo; done; } | tee $O/pmc_c5.txt
for c in bench c3 c5; do f=$(ls -t $O/prof_$c/*/*kernel_stats.csv | head -1); echo "== $c $f"; head -8 "$f" | cut -c1-200; done
timeout 300 python tools/khorn_perf.py 2>&1 | grep "^{" | tee $O/khorn_perf.json
