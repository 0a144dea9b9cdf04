#!/bin/bash
# One GPU-box round: GPU tests, smoke, headline bench, rocprofv3 kernel stats (+ optional PMC passes).
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh [pmc]'
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_1m.log 2>&1; tail -1 gpurun_out/bench_1m.log | cut -c1-600
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
if [ "$1" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- \
        python $GRAFT_REPO_ROOT/tools/knn_perf.py 1000000 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_umap_$c -- \
        python $GRAFT_REPO_ROOT/tools/umap_perf.py 1000000 > $GRAFT_REPO_ROOT/gpurun_out/pmc_umap_$c.log 2>&1
  done
fi
cd $GRAFT_REPO_ROOT; f=$(ls -t gpurun_out/prof_bench/*/*kernel_stats.csv | head -1); head -6 "$f" | cut -c1-160
