import sys, torch, json, time
sys.path.insert(0, "/root/repo")
from tests.conftest import gmm
from torchdr_amd.distance import FaissConfig, pairwise_distances
def recall(I, Ie):
    return float((I[:, :, None] == Ie[:, None, :]).any(2).float().mean())
for (n, d, scale, noise) in ((40000, 24, 1.0, 0.3), (40000, 24, 2.0, 0.0), (200000, 64, 2.0, 0.0)):
    g = torch.Generator().manual_seed(1)
    X = (gmm(n, d, scale, seed=5) + noise * torch.randn(n, d, generator=g)).cuda()
    Ce, Ie = pairwise_distances(X, metric="sqeuclidean", k=15, exclude_diag=True, return_indices=True)
    for nlist in (128, 512, 2048):
        out = {}
        for nprobe in (1, 2, 4, 8, 16, 32):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            C, I = pairwise_distances(X, metric="sqeuclidean", k=15, exclude_diag=True, return_indices=True, backend=FaissConfig(index_type="IVF", nlist=nlist, nprobe=nprobe))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out[nprobe] = (round(recall(I[::7].cpu(), Ie[::7].cpu()), 4), round(dt * 1e3, 1))
        print(json.dumps({"n": n, "d": d, "scale": scale, "noise": noise, "nlist": nlist, "recall,ms": out}), flush=True)
