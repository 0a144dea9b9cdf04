#!/bin/bash
export TMPDIR=/tmp
for g in 16 17 19; do
TDR_GEOM=$g timeout 600 python - <<'PY' 2>&1 | tail -1
import os, sys, json, io, contextlib, runpy
from torchdr_amd.neighbor_embedding import umap as U
U.SCHED_GEOM = int(os.environ["TDR_GEOM"])
sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-knn-variants", "--no-configs"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print("SCHED_GEOM", U.SCHED_GEOM, "ms_per_step", round(d["ms_per_step"], 2), "loop", d["phases_ms"]["loop"], "grad+step ms", round(d["roofline"]["grad_passes_ms"], 5))
PY
done
