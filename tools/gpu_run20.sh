#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/lv.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from tests.conftest import gmm
import torchdr_amd as t
X = gmm(1_000_000, 128, 2.0).cuda()
t.LargeVis(perplexity=5, max_iter=100, random_state=0).fit_transform(X)
torch.cuda.synchronize()
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lv -- python /tmp/lv.py > $GRAFT_REPO_ROOT/gpurun_out/prof_lv.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls -t gpurun_out/prof_lv/*/*kernel_stats.csv | head -1); head -8 "$f" | cut -c1-170
