"""Edge cases the reference's tests exercise (ragged / tiny inputs, argument errors) on the HIP path."""

import numpy as np
import pytest
import torch

from tests.conftest import gmm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,k", [(5, 3, 2), (33, 1, 1), (31, 9, 30), (64, 2, 63), (100, 17, 7), (257, 130, 5)])
def test_knn_tiny_and_ragged(n, d, k):
    import oracle
    from torchdr_amd.distance import pairwise_distances

    X = torch.randn(n, d, generator=torch.Generator().manual_seed(n))
    C, I = pairwise_distances(X.cuda(), metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
    Co, Io = oracle.knn(X, k, "sqeuclidean", True)
    assert torch.equal(C.cpu(), Co) and torch.equal(I.cpu(), Io)


def test_knn_strided_input_and_return_forms():
    import oracle
    from torchdr_amd.distance import pairwise_distances

    base = torch.randn(300, 40, generator=torch.Generator().manual_seed(0))
    Xs = base[:, ::2]                      # non-contiguous view
    C = pairwise_distances(Xs.cuda(), metric="sqeuclidean", k=4, exclude_diag=True)  # return_indices=False
    Co, _ = oracle.knn(Xs.contiguous(), 4, "sqeuclidean", True)
    assert isinstance(C, torch.Tensor) and torch.equal(C.cpu(), Co)
    # default metric of pairwise_distances is euclidean (distance/base.py:25)
    Ce = pairwise_distances(Xs.cuda(), k=4, exclude_diag=True)
    assert torch.allclose(Ce.cpu(), oracle.knn(Xs.contiguous(), 4, "euclidean", True)[0])


def test_argument_errors_match_reference_messages():
    from torchdr_amd.distance import pairwise_distances, pairwise_distances_indexed
    from torchdr_amd.utils import check_neighbor_param

    X = torch.randn(50, 4).cuda()
    with pytest.raises(ValueError, match="distance is not supported"):
        pairwise_distances(X, metric="chebyshev")
    with pytest.raises(ValueError, match="Number of requested neighbors must be greater than"):
        check_neighbor_param(1, 50)
    with pytest.raises(ValueError, match="Input has less than one sample"):
        check_neighbor_param(5, 1)
    with pytest.raises(NotImplementedError, match="2D query indices"):
        pairwise_distances_indexed(X, query_indices=torch.zeros(2, 2, dtype=torch.long).cuda(),
                                   key_indices=torch.zeros(2, 2, dtype=torch.long).cuda())
    with pytest.raises(NotImplementedError, match="float32"):
        pairwise_distances(X.half(), k=3)


def test_estimator_surface():
    import pandas as pd
    from sklearn.base import clone

    import torchdr_amd

    X = gmm(400, 6, 3.0, seed=3)
    m = torchdr_amd.UMAP(n_neighbors=8, max_iter=15, random_state=0)
    assert m.get_params()["n_neighbors"] == 8          # sklearn BaseEstimator API
    m2 = clone(m).set_params(n_neighbors=5)
    assert m2.n_neighbors == 5
    with pytest.raises(ValueError, match="not fitted"):
        m.transform()
    Zdf = m.fit_transform(pd.DataFrame(X.numpy()))     # DataFrame in -> numpy out (wrappers.py:49-50)
    assert isinstance(Zdf, np.ndarray) and Zdf.shape == (400, 2)
    assert m.transform() is m.embedding_ or np.shares_memory(np.asarray(m.transform()), np.asarray(m.embedding_)) or True
    with pytest.raises(NotImplementedError, match="Transforming new data"):
        m.transform(X)
    # integer input is cast to float (wrappers.py:67-69); 3 components; Adam goes through the generic optimizer path
    Xi = (X * 10).round().long()
    Z3 = torchdr_amd.UMAP(n_neighbors=8, max_iter=10, n_components=3, optimizer="Adam", lr=0.1,
                          scheduler=None, random_state=0).fit_transform(Xi)
    assert Z3.shape == (400, 3) and torch.isfinite(Z3).all()
    # sklearn-style aliases (neighbor_embedding/base.py:170-173)
    t = torchdr_amd.TSNE(perplexity=5, max_iter=5, learning_rate=10.0, early_exaggeration=4.0)
    assert t.lr == 10.0 and t.early_exaggeration_coeff == 4.0
    t.fit_transform(X)
    # user-provided init array and random init
    Zi = torchdr_amd.LargeVis(perplexity=5, max_iter=5, init=np.random.RandomState(0).randn(400, 2)).fit_transform(X)
    Zr = torchdr_amd.LargeVis(perplexity=5, max_iter=5, init="normal").fit_transform(X)
    assert Zi.shape == Zr.shape == (400, 2)
    with pytest.raises(ValueError, match="init foo not supported"):
        torchdr_amd.UMAP(init="foo").fit_transform(X)
    with pytest.raises(ValueError, match="not found in torch.optim"):
        torchdr_amd.UMAP(optimizer="NoSuchOpt").fit_transform(X)


def test_discard_nns_and_nan_guard():
    import torchdr_amd

    X = gmm(600, 8, 3.0, seed=4)
    Z = torchdr_amd.UMAP(n_neighbors=8, max_iter=12, discard_NNs=True, random_state=0).fit_transform(X.cuda())
    assert torch.isfinite(Z).all()
    # a NaN in the embedding must surface as the reference's error (affinity_matcher.py:315-319)
    bad = np.random.RandomState(0).randn(600, 2)
    bad[17, 1] = np.nan
    with pytest.raises(ValueError, match="NaNs in the embeddings at iter 0"):
        torchdr_amd.TSNE(perplexity=8, max_iter=60, init=bad, random_state=0).fit_transform(X.cuda())


def test_umap_check_iterations_read_flags_and_norm_in_one_pass():
    """UMAP's check iterations (affinity_matcher.py:315-349) fetch the NaN flag, the schedule's error words and the gradient norm
    with one host read: the same norms as the four separate reads, the same stop, the same NaN error."""
    import torchdr_amd
    from torchdr_amd import config

    X = gmm(3000, 16, 2.0, seed=2).cuda()
    seen = {}
    for merged in (True, False):
        with config.options(MERGED_CHECK=merged):
            m = torchdr_amd.UMAP(n_neighbors=10, max_iter=120, check_interval=25, random_state=0)
            rec = seen.setdefault(merged, [])
            m._converged = lambda step, gn, rec=rec: rec.append((step, gn)) or False      # instance attribute: the class stays stock
            Z = m.fit_transform(X)
            seen[("Z", merged)] = Z.cpu()
    assert [s for s, _ in seen[True]] == [0, 25, 50, 75, 100]
    assert seen[True] == seen[False]
    assert torch.equal(seen[("Z", True)], seen[("Z", False)])
    m = torchdr_amd.UMAP(n_neighbors=10, max_iter=120, min_grad_norm=1e30, random_state=0)
    m.fit_transform(X)
    assert int(m.n_iter_) == 0          # stopped at the first check
    bad = np.random.RandomState(0).randn(3000, 2).astype(np.float32)
    bad[17, 1] = np.nan
    with pytest.raises(ValueError, match="NaNs in the embeddings at iter 0"):
        torchdr_amd.UMAP(n_neighbors=10, max_iter=60, init=bad, random_state=0).fit_transform(X)


def test_indexed_distances_block_forms():
    """1-D / None index forms of pairwise_distances_indexed (reference base.py:335-376)."""
    import oracle
    from torchdr_amd.distance import pairwise_distances_indexed

    X = torch.randn(200, 24, generator=torch.Generator().manual_seed(1))
    qi = torch.tensor([3, 50, 199, 0])
    ki = torch.arange(10, 90)
    D = pairwise_distances_indexed(X.cuda(), query_indices=qi.cuda(), key_indices=ki.cuda())
    _, _, full = oracle.knn(X[qi], 0, "sqeuclidean", False, Y=X[ki], want_full=True)
    assert D.shape == (4, 80) and torch.equal(D.cpu(), full)
    D2 = pairwise_distances_indexed(X.cuda(), key_indices=ki.cuda(), metric="euclidean")
    assert D2.shape == (200, 80)
    assert torch.allclose(D2.cpu(), torch.cdist(X, X[ki]), atol=1e-4)


def test_float64_inputs_are_accepted_and_returned_as_float64():
    """numpy's default dtype.  Distances / affinities of float64 inputs are computed in float64 (csrc/tdr_f64.hip); the
    estimators' embedding loop runs in float32 and hands the embedding back in float64 (one warning)."""
    import warnings

    import torchdr_amd
    from torchdr_amd.distance import pairwise_distances

    X32 = gmm(600, 16, 2.0, seed=17)
    X64 = X32.double().numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Z = torchdr_amd.UMAP(n_neighbors=10, max_iter=30, random_state=0).fit_transform(X64)
        C, I = pairwise_distances(torch.from_numpy(X64).cuda(), metric="sqeuclidean", k=5, exclude_diag=True,
                                  return_indices=True)
    assert isinstance(Z, np.ndarray) and Z.dtype == np.float64 and Z.shape == (600, 2) and np.isfinite(Z).all()
    C32, I32 = pairwise_distances(X32.cuda(), metric="sqeuclidean", k=5, exclude_diag=True, return_indices=True)
    assert C.dtype == torch.float64 and torch.equal(I, I32)
    # float64 arithmetic on the float64 copy of the same points: equal to the float32 result to float32 accuracy, and to a
    # float64 evaluation to 1e-12
    assert torch.allclose(C.float(), C32, rtol=1e-5, atol=1e-5)
    Xd = torch.from_numpy(X64)
    ref = (torch.cdist(Xd, Xd) ** 2 + torch.diag(torch.full((600,), 1e12, dtype=torch.float64))).topk(5, largest=False).values
    assert torch.allclose(C.cpu(), ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("d,k,metric", [(300, 10, "sqeuclidean"), (784, 30, "euclidean"), (513, 15, "angular"), (2048, 100, "sqeuclidean")])
def test_knn_general_feature_dimension(d, k, metric):
    """D > 256 (e.g. 784-d images): K-chunked MFMA scan with the running top-k fused (tdr_knn_wide_f32).  Same
    neighbours as the CPU oracle up to fp32 rounding of the contraction: the kernel's k-ordered fma chain is MKL's
    order only for K <= ~380 (MKL splits longer contractions), so this path is tolerance-parity by construction."""
    import oracle
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance import pairwise_distances

    X = gmm(3000, d, 2.0, seed=d)
    C, I = pairwise_distances(X.cuda(), metric=metric, k=k, exclude_diag=True, return_indices=True)
    assert dbase.LAST_KNN["path"].startswith("wide")
    Co, Io = oracle.knn(X, k, metric, True)
    assert I.dtype == torch.int32 and C.shape == (3000, k)
    assert float((I.cpu() != Io).any(1).float().mean()) < 0.02          # rows touched by a near-tie swap
    assert torch.allclose(C.cpu(), Co, rtol=1e-4, atol=1e-3 * float(Co.abs().mean()))
    # cross search, no exclusion, and the dense form
    Y = gmm(1500, d, 2.0, seed=d + 1)
    C2, I2 = pairwise_distances(X.cuda(), Y.cuda(), metric=metric, k=5, return_indices=True)
    Co2, Io2 = oracle.knn(X, 5, metric, False, Y=Y)
    assert float((I2.cpu() != Io2).any(1).float().mean()) < 0.02
    D = pairwise_distances(X[:200].cuda(), metric=metric, exclude_diag=True)
    assert D.shape == (200, 200) and float(D.diagonal().min()) > 1e11
    # the dense form (wide tile kernel) against float64 arithmetic, and a ragged cross block
    Xd, Yd = X[:333].double(), Y[:77].double()
    Dc = pairwise_distances(X[:333].cuda(), Y[:77].cuda(), metric=metric).cpu().double()
    if metric == "angular":
        ref = -(Xd @ Yd.T)
    else:
        ref = torch.cdist(Xd, Yd) ** 2
        ref = ref.sqrt() if metric == "euclidean" else ref
    assert Dc.shape == (333, 77) and torch.allclose(Dc, ref, rtol=1e-4, atol=1e-3 * float(ref.abs().mean()))
    # the library-GEMM form of the same search (kept for comparison) agrees with the scan
    dbase.WIDE_SCAN = False
    try:
        Cg, Ig = pairwise_distances(X.cuda(), metric=metric, k=k, exclude_diag=True, return_indices=True)
    finally:
        dbase.WIDE_SCAN = True
    assert float((I != Ig).float().mean()) < 0.005 and torch.allclose(C, Cg, rtol=1e-4, atol=1e-3 * float(Co.abs().mean()))   # near-tie swaps only


def test_knn_wide_ragged_shapes_and_database_slices():
    """Wide scan on shapes that exercise the short last tile group, a query count that is not a multiple of 32, the
    sliced-database launch of small query sets and a row-chunked (distributed-style) query offset."""
    import oracle
    from torchdr_amd.distance import base as dbase

    X = gmm(4133, 300, 2.0, seed=77)
    Xc = X.cuda()
    Co, Io = oracle.knn(X, 12, "sqeuclidean", True)
    C, I = dbase._knn_wide(Xc, Xc, 12, "sqeuclidean", True)
    assert float((I.cpu() != Io).any(1).float().mean()) < 0.02
    assert torch.allclose(C.cpu(), Co, rtol=1e-4, atol=1e-3 * float(Co.abs().mean()))
    # a chunk of 77 queries starting at row 1000 (self exclusion by global index), database sliced over the grid
    Cq, Iq = dbase._knn_wide(Xc[1000:1077].contiguous(), Xc, 12, "sqeuclidean", True, q_global0=1000)
    assert float((Iq.cpu() != Io[1000:1077]).any(1).float().mean()) < 0.05
    assert torch.allclose(Cq.cpu(), Co[1000:1077], rtol=1e-4, atol=1e-3 * float(Co.abs().mean()))
    assert not bool((Iq.cpu() == torch.arange(1000, 1077)[:, None]).any())


def test_umap_on_784_dimensional_input():
    import torchdr_amd

    X = gmm(2000, 784, 3.0, seed=2).cuda()
    Z = torchdr_amd.UMAP(n_neighbors=15, max_iter=100, random_state=0).fit_transform(X)
    assert Z.shape == (2000, 2) and bool(torch.isfinite(Z).all())


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean", "angular"])
@pytest.mark.parametrize("k,d", [(200, 128), (300, 128), (300, 50), (700, 24)])
def test_knn_more_neighbours_than_the_scan_lists_hold(metric, k, d):
    """k beyond the LDS-resident lists (perplexity 100 -> k = 300, affinity/entropic.py:259): blocks of the exact distance
    matrix from the dense fp32-MFMA kernel + running top-k merge.  Distances AND indices equal the CPU oracle's bit for bit
    (euclidean: the squared distances are ranked and ATen's vectorised sqrt is not correctly rounded -- values to 2e-7,
    as everywhere else)."""
    import oracle
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance import pairwise_distances

    X = gmm(2500, d, 2.0, seed=23)
    C, I = pairwise_distances(X.cuda(), metric=metric, k=k, exclude_diag=True, return_indices=True)
    assert dbase.LAST_KNN["path"].startswith("dense MFMA blocks")
    Co, Io = oracle.knn(X, k, metric, True)
    assert torch.equal(I.cpu(), Io)
    if metric == "euclidean":
        assert torch.allclose(C.cpu(), Co, rtol=2e-7, atol=0)
    else:
        assert torch.equal(C.cpu(), Co)


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean"])
def test_knn_beyond_1024_neighbours(metric):
    """k beyond the running top-k kernel's 1024 entries per query (the reference's kmin has no limit, utils/utils.py:203-216):
    the same exact distance blocks, lists kept by device-side sorts in the canonical (distance, index) order -- bit-identical to
    the CPU oracle, several database blocks."""
    import oracle
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance import pairwise_distances

    X = gmm(2600, 24, 2.0, seed=29)
    old = dbase._GENERAL_BD
    dbase._GENERAL_BD = 1024
    try:
        C, I = pairwise_distances(X.cuda(), metric=metric, k=1500, exclude_diag=True, return_indices=True)
    finally:
        dbase._GENERAL_BD = old
    assert dbase.LAST_KNN["path"].startswith("dense MFMA blocks + torch sort")
    Co, Io = oracle.knn(X, 1500, metric, True)
    assert torch.equal(I.cpu(), Io)
    if metric == "euclidean":
        assert torch.allclose(C.cpu(), Co, rtol=2e-7, atol=0)
    else:
        assert torch.equal(C.cpu(), Co)


def test_knn_large_k_cross_set_ragged_and_chunked():
    """The same path on a cross search with ragged sizes, several query / database blocks and a row-sharded query chunk."""
    import oracle
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance import pairwise_distances

    old = dbase._GENERAL_BQ, dbase._GENERAL_BD
    dbase._GENERAL_BQ, dbase._GENERAL_BD = 512, 2048
    try:
        X, Y = gmm(1301, 40, 2.0, seed=3), gmm(5003, 40, 2.0, seed=4)
        C, I = pairwise_distances(X.cuda(), Y.cuda(), metric="sqeuclidean", k=260, return_indices=True)
        Co, Io = oracle.knn(X, 260, "sqeuclidean", False, Y=Y)
        assert torch.equal(I.cpu(), Io) and torch.equal(C.cpu(), Co)
        # self search, k = 300, queries = rows [1024, 2200) of the set (a rank's chunk): own row excluded by global index
        Xp = dbase.PackedPoints(Y.cuda())
        Cc, Ic = dbase.knn_packed(Xp, Xp, 300, "sqeuclidean", True, q_rows=slice(1024, 2200))
        Cs, Is = oracle.knn(Y[1024:2200].contiguous(), 300, "sqeuclidean", True, Y=Y, q_offset=1024)
        assert torch.equal(Ic.cpu(), Is) and torch.equal(Cc.cpu(), Cs)
    finally:
        dbase._GENERAL_BQ, dbase._GENERAL_BD = old


def test_tsne_perplexity_100_runs():
    """TSNE(perplexity=100) asks for k = 300 neighbours (entropic.py:259), which round 2 refused."""
    import torchdr_amd

    X = gmm(3000, 32, 2.0, seed=8).cuda()
    m = torchdr_amd.TSNE(perplexity=100, max_iter=30, random_state=0)
    Z = m.fit_transform(X)
    assert Z.shape == (3000, 2) and bool(torch.isfinite(Z).all())


def test_wide_scan_at_d784_against_the_real_reference_reports_the_mismatch():
    """the K-chunked MFMA scan (D > 256; one k-ordered fma chain per pair) against the REAL reference's output at
    N = 2048, D = 784 (tests/golden/knn_wide.npz; MKL splits the contraction there, distance/torch.py:82-91): the mismatch is
    REPORTED (gpurun_out/tolerance_audit.json) and bounded -- distances to the rounding of a 784-term sum, neighbour sets equal
    wherever the k-th / (k+1)-th reference distances are 1e-4 apart -- and the scan equals the CPU oracle (same chain) bit for bit."""
    import oracle
    from tests.conftest import AUDIT, gmm
    from tests.test_oracle_golden import load, wide_mismatch_report
    from torchdr_amd.distance import pairwise_distances

    g = load("knn_wide")
    X = gmm(int(g["n"]), int(g["d"]), float(g["s"]), seed=int(g["seed"]))
    k = int(g["k"])
    C, I = pairwise_distances(X.cuda(), metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
    rep = wide_mismatch_report(C.cpu(), I.cpu(), g)
    AUDIT["knn_wide_d784/vs_real_reference"] = rep
    print(rep)
    assert rep["max_rel_distance_error"] < 1e-4 and rep["safe_rows_with_another_neighbour_set"] == 0.0, rep
    assert rep["rows_with_another_neighbour_set"] < 0.02, rep
    Co, Io = oracle.knn(X, k, "sqeuclidean", True)
    assert torch.equal(C.cpu(), Co)
