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


def test_ivf_nlist_beyond_the_index_builder_is_clamped():
    """FaissConfig(nlist=8192) (the reference's docs recommend 4096..16384 for large sets) used to reach the table
    kernel with more clusters than it builds; it is served with the builder's 4096 lists."""
    from torchdr_amd.distance import FaissConfig, pairwise_distances
    from torchdr_amd.distance import base as dbase

    n, k = 300_000, 10
    X = gmm(n, 16, 2.0, seed=6).cuda()
    C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True,
                              backend=FaissConfig(index_type="IVF", nlist=8192, nprobe=16))
    assert "nlist=4096" in dbase.LAST_KNN["path"]
    assert bool((I >= 0).all()) and bool(torch.isfinite(C).all())


def test_ivf_rows_with_too_few_candidates_are_searched_exactly():
    """nprobe = 1 on many small lists leaves rows with fewer than k candidates; Faiss pads them with -1, which
    the affinity / symmetrisation stages would index the embedding with.  Such rows are searched exactly: no -1, no inf,
    and they equal the exact search's rows."""
    from torchdr_amd.distance import FaissConfig, pairwise_distances

    n, k = 30000, 40
    X = gmm(n, 8, 1.0, seed=2).cuda()
    Ce, Ie = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
    C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True,
                              backend=FaissConfig(index_type="IVF", nlist=468, nprobe=1))    # ~64 rows per list < k + 1 for many
    assert bool((I >= 0).all()) and bool(torch.isfinite(C).all())
    assert bool((C[:, 1:] >= C[:, :-1]).all())
    import torchdr_amd

    Z = torchdr_amd.UMAP(n_neighbors=k, max_iter=30, random_state=0, backend=FaissConfig(index_type="IVF", nlist=468, nprobe=1)).fit_transform(X)
    assert bool(torch.isfinite(Z).all())


def test_ivf_search_at_ten_million_points():
    """SURVEY 8f.4 motivates the approximate search by N >= 10M (reference: IVF nlist 16384 nprobe 81, 54.7 s at 99.9 % recall on
    a B200, BENCHMARK_RESULTS.md:35).  N = 10M x 128 on one MI355X: 4096 lists (the builder's maximum), recall against 2048
    sampled rows searched exactly (one-stage fp32 kernel against all 10M points); every returned distance exact."""
    import time

    from torchdr_amd.distance import FaissConfig, pairwise_distances
    from torchdr_amd.distance import base as dbase

    n, d, k = 10_000_000, 128, 15
    g = torch.Generator().manual_seed(42)
    centers = torch.randn(1000, d, generator=g) * 2.0
    X = torch.empty((n, d), dtype=torch.float32, device="cuda")
    for s in range(0, n, 1_000_000):       # generated in slices: the host never holds the 5 GB block
        lab = torch.arange(s, s + 1_000_000) % 1000
        X[s:s + 1_000_000] = (centers[lab] + 0.5 * torch.randn(1_000_000, d, generator=g)).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True,
                              backend=FaissConfig(index_type="IVF", nlist=4096, nprobe=8))
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    assert dbase.LAST_KNN["path"].startswith("ivf") and bool((I >= 0).all())
    rows = torch.randint(0, n, (2048,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    old = dbase.SCREEN_MODE
    dbase.SCREEN_MODE = "0"
    try:
        Ce, Ie = pairwise_distances(X[rows].contiguous(), X, metric="sqeuclidean", k=k + 1, return_indices=True)
    finally:
        dbase.SCREEN_MODE = old
    keep = Ie != rows[:, None].int()
    ok = keep.sum(1) == k
    Ie_k, Ce_k = Ie[ok][keep[ok]].reshape(-1, k), Ce[ok][keep[ok]].reshape(-1, k)
    rec = recall(I[rows][ok].cpu(), Ie_k.cpu())
    print({"ivf_10m_sec": round(sec, 2), "recall": round(rec, 4)})
    assert rec > 0.99, rec
    hit = I[rows][ok] == Ie_k           # where the same neighbour sits in the same slot, the distance is the exact one
    assert torch.equal(C[rows][ok][hit], Ce_k[hit])
    assert sec < 30.0


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean"])
def test_ivf_equals_the_cpu_restatement_on_the_same_index(metric):
    """the approximate search pinned to a CPU restatement (oracle/ref_torch.py:ivf_search -- probe rule +
    exact top-k over the probed lists with the reference's arithmetic + exact fallback for short rows).  The restatement is
    fed the HIP index's own tables (sorted order, list of every tile, centre distances); indices AND distances must be equal
    bit for bit at nprobe = 1, 8 and nlist."""
    from oracle import ref_torch as R
    from torchdr_amd.distance import base as dbase

    n, d, k, nlist = 20000, 24, 10, 128
    X = (gmm(n, d, 1.0, seed=11) + 0.3 * torch.randn(n, d, generator=torch.Generator().manual_seed(2))).cuda()
    Yp = dbase.PackedPoints(X)
    for nprobe in (1, 8, nlist):
        C, I = dbase._knn_ivf(Yp, k, metric, True, nlist, nprobe)
        ci = Yp._ivf_index[nlist]
        assert ci.n_clusters == nlist
        Co, Io, short = R.ivf_search(X.cpu(), ci.row_map, ci.tile_cluster, ci.dist.view(nlist, nlist), nprobe, k, metric, True)
        # rows the kernel could not fill (fewer than k candidates in the probed lists) and ONLY those were searched exactly
        assert dbase.LAST_KNN["flagged"] == int(short.sum()), (nprobe, dbase.LAST_KNN["flagged"], int(short.sum()))
        assert torch.equal(I.cpu(), Io), (nprobe, float((I.cpu() != Io).float().mean()))
        assert torch.equal(C.cpu(), Co), nprobe
