#!/bin/bash
# Round-4 closing measurement pass on one MI355X box:  gpurun --timeout 3000 -- 'bash tools/gpu_round4_final.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final4
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/tests.log; tail -3 $O/tests.log
cp gpurun_out/tolerance_audit.json $O/tolerance_audit.json 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 5 --warmup 1 > $O/bench.log 2>&1; grep "^{" $O/bench.log | cut -c1-300
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-knn-variants --no-configs > $O/bench_prof.log 2>&1
cd $R
f=$(ls -t $O/prof_bench/*/*kernel_stats.csv | head -1); echo "== $f"; head -12 "$f" | cut -c1-220; cp "$f" $O/bench_kernel_stats.csv
bash tools/pmc_bench.sh r04 > $O/pmc_bench.log 2>&1; tail -12 $O/pmc_bench.log
bash tools/pmc_build2.sh 0 > $O/pmc_build2.log 2>&1; tail -7 $O/pmc_build2.log | cut -c1-600
