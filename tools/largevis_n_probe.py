"""GPU probe: LargeVis / InfoTSNE / TSNE fits over N (bench generator, D = 128): ms per fit, to spot sizes that fall off a path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import gmm
import torchdr_amd as t
from torchdr_amd.distance import base as dbase

for n in (100_000, 300_000, 500_000, 700_000, 1_000_000):
    X = gmm(n, 128, 2.0).cuda()
    for name, make in (("LargeVis 100 it", lambda: t.LargeVis(perplexity=5, max_iter=100, random_state=0)),
                       ("InfoTSNE 30 it", lambda: t.InfoTSNE(perplexity=10, max_iter=30, random_state=0))):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            Z = make().fit_transform(X)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"N={n:8d} {name:16s} {dt * 1e3:8.1f} ms  knn={dbase.LAST_KNN.get('path')}", flush=True)
    del X
