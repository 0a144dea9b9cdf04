#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_knn_screen_gpu.py tests/test_knn_gpu.py tests/test_knn_ivf_gpu.py tests/test_umap_sched_gpu.py tests/test_edge_cases_gpu.py -q -x 2>&1 | tail -8 > gpurun_out/r4g_tests.log; tail -4 gpurun_out/r4g_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > gpurun_out/r4g_bench.log 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r4g_bench.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['ms_per_step'], d['phases_ms']['knn'], d['roofline']['schedule_build_ms_per_iteration']); print(d['knn_context']); print({k:d['knn_uniform'][k] for k in ('sec','tier','frac_of_f16_peak')})
else:
    print(open('gpurun_out/r4g_bench.log').read()[-2000:])
PY
timeout 200 python tools/sched_build2_perf.py > gpurun_out/r4g_build2.log 2>&1; grep "^{" gpurun_out/r4g_build2.log | cut -c1-220
