"""GPU probe: the UMAP fit on one GPU with the Python loop (default there) against the C loop object with HIP-graph replays
(LOOP_RUNNER = True), over N:   python tools/loop_runner_probe.py [sizes, e.g. 100000,300000,1000000]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import torchdr_amd
from tests.conftest import gmm
from torchdr_amd import config
from torchdr_amd.utils import phases

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "100000,300000,1000000").split(",")]
for n in sizes:
    X = gmm(n, 128, 2.0).cuda()
    out = {"n": n}
    for name, runner in (("python_loop", False), ("c_loop_graphs", True)):
        best = None
        for rep in range(3):
            with config.options(LOOP_RUNNER=runner):
                torch.cuda.synchronize()
                phases.start()
                t0 = time.perf_counter()
                Z = torchdr_amd.UMAP(n_neighbors=30, max_iter=1000, random_state=0).fit_transform(X)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) * 1e3
                ph = phases.stop()
            if rep and (best is None or wall < best[0]):
                best = (wall, ph.get("loop"))
        out[name] = {"fit_ms": round(best[0], 2), "loop_ms": round(best[1], 2)}
        out[name + "_finite"] = bool(torch.isfinite(Z).all())
    print(json.dumps(out), flush=True)
