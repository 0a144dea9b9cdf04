import sys, torch, json, time
sys.path.insert(0, "/root/repo")
from tests.conftest import gmm
from torchdr_amd.distance import FaissConfig, pairwise_distances
def recall(I, Ie):
    return float((I[:, :, None] == Ie[:, None, :]).any(2).float().mean())
n, d, k = 1_000_000, 128, 30
X = gmm(n, d, 2.0).cuda()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Ce, Ie = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
    torch.cuda.synchronize(); te = time.perf_counter() - t0
print(json.dumps({"exact_ms": round(te * 1e3, 1)}), flush=True)
sel = torch.arange(0, n, 53)
for nlist in (1024, 4096):
    out = {}
    for nprobe in (1, 4, 16, 40):
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True, backend=FaissConfig(index_type="IVF", nlist=nlist, nprobe=nprobe))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[nprobe] = (round(recall(I[sel].cpu(), Ie[sel].cpu()), 4), round(dt * 1e3, 1))
    print(json.dumps({"n": n, "d": d, "k": k, "nlist": nlist, "recall,ms (index cached)": out}), flush=True)
