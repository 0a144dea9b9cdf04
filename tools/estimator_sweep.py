"""GPU smoke sweep: every estimator at a size where the kNN search prunes (adaptive index, cluster-ordered loop), a few iterations each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import gmm
import torchdr_amd as t
from torchdr_amd.distance import base as dbase

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
for scale in (2.0, 1.0, 0.0):
    X = gmm(n, 64, scale).cuda()
    for name, make in (("UMAP", lambda: t.UMAP(n_neighbors=15, max_iter=40, random_state=0)),
                       ("UMAP nc=3 euclidean", lambda: t.UMAP(n_neighbors=10, max_iter=20, n_components=3, metric="euclidean", random_state=0)),
                       ("LargeVis", lambda: t.LargeVis(perplexity=8, max_iter=20, random_state=0)),
                       ("InfoTSNE", lambda: t.InfoTSNE(perplexity=8, max_iter=10, random_state=0)),
                       ("TSNE", lambda: t.TSNE(perplexity=8, max_iter=6, random_state=0)),
                       ("SNE", lambda: t.SNE(perplexity=8, max_iter=4, random_state=0)),
                       ("PACMAP", lambda: t.PACMAP(n_neighbors=10, max_iter=20, random_state=0))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Z = make().fit_transform(X)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ok = bool(torch.isfinite(Z).all()) and Z.shape[0] == n
        print(f"scale {scale} {name:22s} {dt * 1e3:8.1f} ms  finite={ok}  knn={dbase.LAST_KNN.get('path')} tile_bounds={dbase.LAST_KNN.get('tile_bounds')} flat_terms={dbase.LAST_KNN.get('flat_terms')}", flush=True)
