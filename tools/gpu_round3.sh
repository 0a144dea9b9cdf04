#!/bin/bash
# Round-3 measurement pass on one MI355X box:  gpurun --timeout 2400 -- 'bash tools/gpu_round3.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/tests_r03c.log; tail -3 gpurun_out/tests_r03c.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 700 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r03c.log 2>&1; grep "^{" gpurun_out/bench_r03c.log | cut -c1-300
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_r03c -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-knn-variants > $R/gpurun_out/bench_prof_r03c.log 2>&1
cd $R
# HBM-side traffic of the production loop kernels (FETCH_SIZE / WRITE_SIZE passes of tools/pmc_sched.sh)
RELABEL=1 timeout 400 bash tools/pmc_sched.sh "4 5" 80 > gpurun_out/pmc_sched_r03.log 2>&1; tail -4 gpurun_out/pmc_sched_r03.log | cut -c1-400
# C3 after the loop moved to cluster order: bench line, kernel stats, counters
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3c -- python $R/tools/config_roofline.py c3 > $R/gpurun_out/roof_c3c.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_c3c_$i -- python $R/tools/config_roofline.py c3 > $R/gpurun_out/pmc_c3c_$i.log 2>&1
done
cd $R
grep "^{" gpurun_out/roof_c3c.log | cut -c1-200
for i in 1 2 3 4; do python tools/pmc_sum.py gpurun_out/pmc_c3c_$i ne_grad_kernel | tr -d '\n'; echo; done | tee gpurun_out/pmc_c3c.txt
