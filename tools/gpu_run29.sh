#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_1m.log 2>&1; tail -1 gpurun_out/bench_1m.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','knn_build_sec')}, d['roofline']['achieved'], d['roofline']['frac'], d['cpu_baseline']['value'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls -t gpurun_out/prof_final/*/*kernel_stats.csv | head -1); head -5 "$f" | cut -c1-150
