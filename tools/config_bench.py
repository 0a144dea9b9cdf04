"""GPU probe: wall time of the BASELINE.json configs other than the headline (parity-test cases, not bench lines)."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.conftest import gmm
import torchdr_amd as t
from torchdr_amd.distance import pairwise_distances

def timed(name, fn, reps=2):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(json.dumps({"config": name, "sec": round(best, 4)}), flush=True)
    return out

which = sys.argv[1:] or ["c1", "c2", "c3", "c5"]
if "c1" in which:
    X = gmm(5000, 50, 2.0).cuda()
    timed("C1 TSNE 5k x 50 perplexity 30, max_iter 2000 (defaults)", lambda: t.TSNE(perplexity=30, random_state=0).fit_transform(X), 1)
if "c2" in which:
    X = gmm(100_000, 128, 2.0).cuda()
    timed("C2 kNN N=100k D=128 k=30 (pairwise_distances)", lambda: pairwise_distances(X, metric="sqeuclidean", k=30, exclude_diag=True, return_indices=True))
    timed("C2 UMAP N=100k D=128 k=30 fit_transform", lambda: t.UMAP(n_neighbors=30, random_state=0).fit_transform(X))
if "c3" in which:
    X = gmm(1_000_000, 128, 2.0).cuda()
    timed("C3 LargeVis N=1M D=128 perplexity 5 (kNN width 15), 500 iters", lambda: t.LargeVis(perplexity=5, max_iter=500, random_state=0).fit_transform(X), 2)
    timed("C3b LargeVis N=1M D=128 perplexity 15 (kNN width 45), 500 iters", lambda: t.LargeVis(perplexity=15, max_iter=500, random_state=0).fit_transform(X), 1)
if "c5" in which:
    n = 200_000
    X = gmm(n, 64, 2.0).cuda()
    sea = t.SymmetricEntropicAffinity(perplexity=30, lr=1e-1, max_iter=100, zero_diag=False, verbose=False)
    timed("C5 SEA duals N=200k D=64 perplexity 30, 100 Adam iterations (matrix-free)", lambda: sea.fit_duals(X), 1)
    print(json.dumps({"sea_n_iter": int(sea.n_iter_)}))
    m = t.TSNEkhorn(perplexity=30, max_iter=20, max_iter_affinity_in=5, init="normal", init_scaling=1.0, lr=1.0,
                    optimizer="SGD", optimizer_kwargs=None, min_grad_norm=1e-30, random_state=0)
    timed("C5 TSNEkhorn N=200k: 5 SEA iterations + 20 training steps (5 Sinkhorn passes + fused force each)", lambda: m.fit_transform(X), 1)
if "c4" in which:
    # C4 is the 8-GPU configuration (N=4M, D=256, k=30); on ONE MI355X it is a capacity / large-N robustness probe
    from torchdr_amd.distance import base as dbase
    n = 4_000_000
    X = gmm(n, 256, 2.0).cuda()
    C, I = timed("C4 kNN N=4M D=256 k=30 on one GPU (pairwise_distances)",
                 lambda: pairwise_distances(X, metric="sqeuclidean", k=30, exclude_diag=True, return_indices=True), 1)
    print(json.dumps({"knn_path": dbase.LAST_KNN.get("path"), "tier": dbase.LAST_KNN.get("tier"), "flagged": int(dbase.LAST_KNN.get("flagged", 0)),
                      "self_returned": bool((I == torch.arange(n, device="cuda", dtype=torch.int32)[:, None]).any()),
                      "sorted": bool((C[:, 1:] >= C[:, :-1]).all())}))
    # sampled parity at full size: 4096 random rows re-searched by the one-stage exact fp32 kernel (k + 1 without
    # exclusion, then the row itself dropped) must give the same neighbours and distances, bit for bit
    rows = torch.randint(0, n, (4096,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    old = dbase.SCREEN_MODE
    dbase.SCREEN_MODE = "0"
    try:
        Ce, Ie = pairwise_distances(X[rows].contiguous(), X, metric="sqeuclidean", k=31, return_indices=True)
    finally:
        dbase.SCREEN_MODE = old
    keep = Ie != rows[:, None].int()
    ok_rows = keep.sum(1) == 30
    Ie30 = Ie[ok_rows][keep[ok_rows]].reshape(-1, 30)
    Ce30 = Ce[ok_rows][keep[ok_rows]].reshape(-1, 30)
    print(json.dumps({"sampled_rows": int(ok_rows.sum()), "indices_equal": bool(torch.equal(Ie30, I[rows][ok_rows])),
                      "distances_equal": bool(torch.equal(Ce30, C[rows][ok_rows]))}))
    del C, I
    Z = timed("C4 UMAP N=4M D=256 k=30, 200 iterations, one GPU", lambda: t.UMAP(n_neighbors=30, max_iter=200, random_state=0).fit_transform(X), 1)
    print(json.dumps({"finite": bool(torch.isfinite(Z).all()), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
