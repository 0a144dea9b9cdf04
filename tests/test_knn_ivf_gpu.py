"""Approximate (IVF-style) kNN on the cluster index -- SURVEY 8f.4, reference distance/faiss.py:331-349.

The reference hands `FaissConfig(index_type="IVF", nlist, nprobe)` to Faiss; here the same object selects
`tdr_knn_ivf_f32`.  What an IVF index promises is checked: every returned pair carries the exact reference distance, recall
grows with nprobe and reaches the exact result when every cluster may be scanned, and callers that pass the config to an
estimator get a working fit."""

import pytest
import torch

from tests.conftest import gmm

pytestmark = pytest.mark.gpu


def recall(I, Ie):
    hit = (I[:, :, None] == Ie[:, None, :]).any(2)
    return float(hit.float().mean())


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean"])
def test_ivf_recall_grows_with_nprobe_and_distances_are_exact(metric):
    from torchdr_amd.distance import FaissConfig, pairwise_distances
    from torchdr_amd.distance import base as dbase

    n, d, k = 40000, 24, 15
    X = (gmm(n, d, 1.0, seed=5) + 0.3 * torch.randn(n, d, generator=torch.Generator().manual_seed(1))).cuda()   # 400 blobs
    Ce, Ie = pairwise_distances(X, metric=metric, k=k, exclude_diag=True, return_indices=True)
    rec = {}
    for nprobe in (1, 4, 16, 512):
        cfg = FaissConfig(index_type="IVF", nlist=512, nprobe=nprobe)
        C, I = pairwise_distances(X, metric=metric, k=k, exclude_diag=True, return_indices=True, backend=cfg)
        assert dbase.LAST_KNN["path"].startswith("ivf") and I.dtype == torch.int32 and C.shape == (n, k)
        assert not bool((I == torch.arange(n, device="cuda", dtype=torch.int32)[:, None]).any())      # self excluded
        found = I >= 0
        assert bool(found[:, 0].all())
        # every returned pair carries the exact distance of that pair, rows ascending
        Xd = X.double()
        rows = torch.arange(0, n, 97, device="cuda")
        ref = ((Xd[rows][:, None, :] - Xd[I[rows].long().clamp(min=0)]) ** 2).sum(-1)
        ref = ref.sqrt() if metric == "euclidean" else ref
        got = C[rows].double()
        m = found[rows]
        assert torch.allclose(got[m], ref[m], rtol=1e-4, atol=1e-4)
        assert bool((C[:, 1:] >= C[:, :-1]).all())
        rec[nprobe] = recall(I.cpu(), Ie.cpu())
    print(rec)
    assert rec[1] < rec[4] <= rec[16] <= rec[512], rec
    assert rec[1] > 0.5 and rec[4] > 0.9 and rec[16] > 0.97, rec
    # nprobe = nlist: nothing is left unvisited except what the exact bound excludes -> the exact result, bit for bit
    assert rec[512] == 1.0
    C, I = pairwise_distances(X, metric=metric, k=k, exclude_diag=True, return_indices=True,
                              backend=FaissConfig(index_type="IVF", nlist=512, nprobe=512))
    assert torch.equal(I, Ie) and torch.equal(C, Ce)


def test_ivfpq_request_and_unsupported_shapes_fall_back_gracefully():
    from torchdr_amd.distance import FaissConfig, pairwise_distances
    from torchdr_amd.distance import base as dbase

    X = gmm(20000, 16, 2.0, seed=3).cuda()
    Ce, Ie = pairwise_distances(X, metric="sqeuclidean", k=10, exclude_diag=True, return_indices=True)
    # IVFPQ: the uncompressed IVF search answers (no product quantisation)
    C, I = pairwise_distances(X, metric="sqeuclidean", k=10, exclude_diag=True, return_indices=True,
                              backend=FaissConfig(index_type="IVFPQ", nlist=64, nprobe=8, M=8))
    assert dbase.LAST_KNN["path"].startswith("ivf") and recall(I.cpu(), Ie.cpu()) > 0.95
    # a cross search, a metric or a size the IVF kernel does not serve: the exact search answers the request
    Y = gmm(5000, 16, 2.0, seed=4).cuda()
    C2, I2 = pairwise_distances(X[:3000], Y, metric="sqeuclidean", k=5, return_indices=True,
                                backend=FaissConfig(index_type="IVF", nlist=64, nprobe=2))
    C3, I3 = pairwise_distances(X[:3000], Y, metric="sqeuclidean", k=5, return_indices=True)
    assert torch.equal(I2, I3) and torch.equal(C2, C3)
    C4, I4 = pairwise_distances(X, metric="angular", k=5, exclude_diag=True, return_indices=True,
                                backend=FaissConfig(index_type="IVF", nlist=64, nprobe=2))
    C5, I5 = pairwise_distances(X, metric="angular", k=5, exclude_diag=True, return_indices=True)
    assert torch.equal(I4, I5)
    with pytest.raises(ValueError, match="Index type.*not supported"):   # reported by the search, as in the reference
        pairwise_distances(X, metric="sqeuclidean", k=5, exclude_diag=True, return_indices=True,
                           backend=FaissConfig(index_type="HNSW"))


def test_umap_with_an_ivf_backend():
    """`backend=FaissConfig(index_type="IVF", ...)` on an estimator (reference: neighbor_embedding/umap.py `backend`)."""
    import torchdr_amd
    from torchdr_amd.distance import FaissConfig
    from torchdr_amd.eval import knn_label_accuracy  # noqa: F401

    n = 20000
    X = gmm(n, 32, 3.0, seed=9).cuda()
    m = torchdr_amd.UMAP(n_neighbors=15, max_iter=150, random_state=0, backend=FaissConfig(index_type="IVF", nlist=64, nprobe=4))
    Z = m.fit_transform(X)
    assert Z.shape == (n, 2) and bool(torch.isfinite(Z).all())
    # blobs stay together: nearest neighbours in the embedding share the blob label far above chance
    labels = (torch.arange(n) % (n // 100)).cuda()
    from torchdr_amd.distance import pairwise_distances
    _, I = pairwise_distances(Z, metric="sqeuclidean", k=5, exclude_diag=True, return_indices=True)
    agree = float((labels[I.long()] == labels[:, None]).float().mean())
    assert agree > 0.8, agree
