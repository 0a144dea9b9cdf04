#!/bin/bash
# round 5 measurement pass on one MI355X box:  gpurun --timeout 3000 -- 'bash tools/gpu_r5_round.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_round; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -v 2>&1 | grep -v "socket.cpp" > $O/tests.log; grep "tests/test_" $O/tests.log | grep -v "PASSED\|SKIPPED" | tail -12 | cut -c1-250; tail -3 $O/tests.log | cut -c1-200
cp gpurun_out/tolerance_audit.json $O/tolerance_audit.json 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py --steps 5 --warmup 1 > $O/bench.log 2>&1; grep "^{" $O/bench.log > $O/bench.json; cut -c1-400 $O/bench.json
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-knn-variants --no-configs > $O/bench_prof.log 2>&1
cd $R
f=$(ls -t $O/prof_bench/*/*kernel_stats.csv | head -1); echo "== $f"; head -14 "$f" | cut -c1-200; cp "$f" $O/bench_kernel_stats.csv
for w in uniform structureless mixture; do timeout 200 python tools/knn_flat_search.py 1000000 $w 2>&1 | grep sec | tail -1; done | tee $O/flat_search.log
timeout 900 bash tools/pmc_flat.sh 1 2>&1 | tail -12
