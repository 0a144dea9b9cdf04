"""Body of __graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against
the oracle (TEST INFRASTRUCTURE -- the only non-test caller of oracle/ besides bench.py's cpu_baseline)."""

import torch


def run_smoke():
    import oracle
    from tests.conftest import gmm
    from torchdr_amd import UMAP
    from torchdr_amd.distance import pairwise_distances

    X = gmm(3000, 64, 2.0, seed=1)
    C, I = pairwise_distances(X.cuda(), metric="sqeuclidean", k=15, exclude_diag=True, return_indices=True)
    Co, Io = oracle.knn(X, 15)
    assert torch.equal(C.cpu(), Co) and torch.equal(I.cpu(), Io), "kNN differs from the oracle"
    Z = UMAP(n_neighbors=15, max_iter=30, random_state=0).fit_transform(X.cuda())
    torch.cuda.synchronize()
    assert Z.shape == (3000, 2) and torch.isfinite(Z).all()
    print("smoke ok: kNN bit-exact vs oracle; UMAP 30 iterations finite")
