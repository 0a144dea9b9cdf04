#!/bin/bash
# Round-6 closing pass on one MI355X:   gpurun --timeout 3000 -- 'bash tools/gpu_r6_round.sh'
#   1. the default bench line (N = 1M; roofline, cpu_baseline, configs C3 / C5, kNN context searches)  -> gpurun_out/r06_bench_1m.json
#   2. rocprofv3 --kernel-trace --stats of the same command without the CPU baseline / variants        -> gpurun_out/r06_bench_1m_kernel_stats.csv
#   3. counters of the pool gradient launch (tools/pmc_pool.sh) and FETCH / WRITE of the whole iteration over the bench command
#      (tools/pmc_bench.sh), separate --pmc passes                                                      -> gpurun_out/r06_pool_pmc.json, r06_umap_pool_pmc.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r06_bench_1m.json 2> gpurun_out/r06_bench_1m.err
tail -c 1600 gpurun_out/r06_bench_1m.json
cd /tmp
rm -rf $R/gpurun_out/prof_r06
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06 -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-knn-variants --no-configs > $R/gpurun_out/r06_bench_prof.log 2>&1
cd $R
f=$(find gpurun_out/prof_r06 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_bench_1m_kernel_stats.csv; head -6 gpurun_out/r06_bench_1m_kernel_stats.csv | cut -c1-160
bash tools/pmc_pool.sh 0 > gpurun_out/r06_pmc_pool.log 2>&1; tail -5 gpurun_out/r06_pmc_pool.log | cut -c1-200
bash tools/pmc_bench.sh r06 > gpurun_out/r06_pmc_bench.log 2>&1; tail -12 gpurun_out/r06_pmc_bench.log | cut -c1-200
