#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_umap_sched_gpu.py tests/test_embed_gpu.py -q -k "umap or fused or sched" 2>&1 | tail -4
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-knn-variants --no-configs > gpurun_out/r4j_bench.log 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r4j_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(d['ms_per_step'], d['phases_ms']['loop'], d['roofline']['grad_passes_ms'], d['roofline']['schedule_build_ms_per_iteration'])
PY
