#!/bin/bash
# loop driver at the headline size: Python iteration per step (default on one GPU) vs the C loop object (plain launches / replayed graphs)
export TMPDIR=/tmp
O=gpurun_out/r4loop; mkdir -p $O
for n in ${SIZES:-1000000}; do
 for l in python c graph; do
  timeout 300 python bench.py --npoints $n --steps 3 --warmup 1 --loop $l --no-cpu-baseline --no-knn-variants --no-configs 2>$O/err_${n}_$l.log | grep "^{" > $O/n${n}_$l.json
  python - <<P
import json
try:
    d=json.load(open("$O/n${n}_$l.json"))
    print($n, "$l", round(d["ms_per_step"],2), {k: round(v,1) for k,v in d["phases_ms"].items() if k in ("knn","loop")})
except Exception as e:
    print($n, "$l", "failed", e)
P
 done
done
