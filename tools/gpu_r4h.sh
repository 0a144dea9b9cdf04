#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_umap_sched_gpu.py tests/test_redzones_gpu.py -q 2>&1 | tail -12 > gpurun_out/r4h_tests.log; tail -8 gpurun_out/r4h_tests.log
for f in 1 0; do
TDR_FUSE=$f timeout 600 python - <<'PY' 2>&1 | tail -2
import os, sys, json, subprocess
from torchdr_amd.neighbor_embedding import umap as U
U.FUSE_STEP = os.environ["TDR_FUSE"] == "1"
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-knn-variants", "--no-configs"]
import runpy, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print("FUSE_STEP", U.FUSE_STEP, "ms_per_step", round(d["ms_per_step"], 2), "loop", d["phases_ms"]["loop"], "grad+step ms", d["roofline"]["grad_passes_ms"], "build/it", d["roofline"]["schedule_build_ms_per_iteration"])
PY
done
