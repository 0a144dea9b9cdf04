#!/bin/bash
# HBM-side traffic of the headline loop's kernels, from the bench command itself (one fit after one warm-up fit):
# separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
#   gpurun --timeout 900 -- 'bash tools/pmc_bench.sh r06'      -> gpurun_out/r06_umap_pool_pmc.json (copy to profiles/)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
mkdir -p $R/gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_bench_$c -- \
      python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-knn-variants --no-configs > $R/gpurun_out/pmc_bench_$c.log 2>&1
done
cd $R
OUT=${2:-${TAG}_umap_pool_pmc.json}
python tools/pmc_iter.py gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE > gpurun_out/$OUT
cat gpurun_out/$OUT | head -40
