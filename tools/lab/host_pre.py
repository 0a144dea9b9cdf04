import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torchdr_amd
from tests.conftest import gmm
import torchdr_amd.utils.wrappers as W
import torchdr_amd.base as B
import torchdr_amd.affinity_matcher as AM
T = {}
def wrap(mod, name, key=None):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T.setdefault(key or name, []).append((time.perf_counter() - t0) * 1e3)
    setattr(mod, name, g)
wrap(W, "to_torch"); wrap(W, "validate_tensor"); wrap(W, "restore_original_format"); wrap(B, "unique_rows"); wrap(B, "as_float32")
for nm in ("_start_pca_prefetch", "on_affinity_computation_start", "_compute_affinity_in", "on_affinity_computation_end", "_init_embedding", "clear_memory", "_run_training_loop"):
    wrap(torchdr_amd.UMAP, nm)
X = gmm(1_000_000, 128, 2.0).cuda()
for r in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = torchdr_amd.UMAP(n_neighbors=30, max_iter=1000, random_state=r)
    t1 = time.perf_counter()
    Z = m.fit_transform(X)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    T.setdefault("ctor", []).append((t1 - t0) * 1e3); T.setdefault("fit_transform", []).append((t2 - t1) * 1e3); T.setdefault("final sync", []).append((t3 - t2) * 1e3)
for k, v in T.items():
    print(f"{k:34s}", [round(x, 3) for x in v[-3:]])
