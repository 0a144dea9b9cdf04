#!/bin/bash
# round-5 size sweep + shape matrix of the unpruned search:  gpurun --timeout 1500 -- 'bash tools/gpu_r5_sizes.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_sizes; mkdir -p $O
timeout 600 python tools/knn_flat_matrix.py 2>$O/matrix.err | tee $O/knn_flat_matrix.jsonl | cut -c1-260
tail -3 $O/matrix.err | cut -c1-300
for n in 100000 300000 500000 700000 1000000 2000000; do
  timeout 300 python bench.py --npoints $n --steps 3 --warmup 1 --no-cpu-baseline --no-knn-variants --no-configs 2>/dev/null | grep "^{" > $O/bench_$n.json
  python -c "
import json; d=json.load(open('$O/bench_$n.json')); print($n, round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['phases_ms'].items() if k in ('knn','loop')}, d.get('knn_path'))"
done
