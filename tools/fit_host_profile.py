"""Host-side profile of one headline fit (cProfile, cumulative): what the Python side spends between the kernels.

    python tools/fit_host_profile.py [N] > gpurun_out/fit_host_profile.txt
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import torchdr_amd
from tests.conftest import gmm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X = gmm(n, 128, 2.0).cuda()
for r in range(2):
    torchdr_amd.UMAP(n_neighbors=30, max_iter=1000, random_state=r).fit_transform(X)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
m = torchdr_amd.UMAP(n_neighbors=30, max_iter=1000, random_state=2)
Z = m.fit_transform(X)
torch.cuda.synchronize()
pr.disable()
print("wall ms (under cProfile):", (time.perf_counter() - t0) * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(70)
