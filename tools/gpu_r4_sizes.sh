#!/bin/bash
# sizes around the headline (same generator), round-4 defaults:  gpurun --timeout 900 -- 'bash tools/gpu_r4_sizes.sh'
export TMPDIR=/tmp
O=gpurun_out/r4sizes; mkdir -p $O
for n in 100000 300000 500000 700000 2000000; do
  timeout 300 python bench.py --npoints $n --steps 3 --warmup 1 --no-cpu-baseline --no-knn-variants --no-configs 2>/dev/null | grep "^{" > $O/n$n.json
  python - <<P
import json
d=json.load(open("$O/n$n.json"))
print($n, round(d["ms_per_step"],2), {k: round(v,1) for k,v in d["phases_ms"].items() if k in ("knn","loop")}, d["knn_path"])
P
done
