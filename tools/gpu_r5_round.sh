#!/bin/bash
# round 5: the whole GPU suite, then the three unpruned searches (threshold scan) with a kernel trace of one of them
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/tests.log; tail -6 $O/tests.log | cut -c1-200
for w in uniform structureless mixture; do timeout 200 python tools/knn_flat_search.py 1000000 $w 2>&1 | grep sec; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_uniform -- python $R/tools/knn_flat_search.py 1000000 uniform > $O/trace_uniform.log 2>&1
f=$(ls -t $O/trace_uniform/*/*kernel_stats.csv | head -1); head -12 "$f" | cut -c1-160
