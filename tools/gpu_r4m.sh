#!/bin/bash
# build-ahead experiment: tests of the scheduled loop + headline bench with BUILD_AHEAD 2 (loop on a high-priority stream) / 1 / 0
export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
timeout 900 python -m pytest tests/test_umap_sched_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/tests.log; tail -4 $O/tests.log
for a in 2 1 0 2 1 0; do
TDR_BUILD_AHEAD=$a timeout 600 python - <<P 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"loop": [0-9.]*\|"grad_passes_ms": [0-9.]*' | tr '\n' ' '; echo " <- BUILD_AHEAD=$a"
import os, json, subprocess, sys
from torchdr_amd.neighbor_embedding import umap as umod
umod.BUILD_AHEAD = int(os.environ["TDR_BUILD_AHEAD"])
sys.argv = ["bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-knn-variants", "--no-configs"]
import runpy
runpy.run_path("bench.py", run_name="__main__")
P
done
