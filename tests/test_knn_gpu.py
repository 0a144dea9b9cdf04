"""K1 parity (GPU): HIP exact kNN vs the C oracle -- bit-exact distances AND indices."""

import pytest
import torch

from tests.conftest import gmm

pytestmark = pytest.mark.gpu


def _run(X, k, metric, exclude, Y=None):
    from torchdr_amd.distance import pairwise_distances

    Xg = X.cuda()
    Yg = None if Y is None else Y.cuda()
    C, I = pairwise_distances(Xg, Yg, metric=metric, k=k, exclude_diag=exclude, return_indices=True)
    torch.cuda.synchronize()
    return C.cpu(), I.cpu()


@pytest.mark.parametrize(
    "n,d,scale,k,metric",
    [
        (2048, 128, 2.0, 30, "sqeuclidean"),
        (2048, 128, 10.0, 30, "sqeuclidean"),   # tie-rich stress variant (reference benchmark scale)
        (3000, 50, 2.0, 90, "sqeuclidean"),     # TSNE-style k = 3 * perplexity, ragged N and D
        (1000, 7, 1.0, 15, "sqeuclidean"),      # D < 8: scalar ATen sum path
        (4100, 32, 0.0, 15, "angular"),
        (2500, 256, 2.0, 30, "sqeuclidean"),
        (5000, 64, 2.0, 45, "sqeuclidean"),
        (40000, 128, 2.0, 30, "sqeuclidean"),   # large enough for every WG slot + db split
    ],
)
def test_knn_bit_exact_vs_oracle(n, d, scale, k, metric):
    import oracle

    X = gmm(n, d, scale)
    C, I = _run(X, k, metric, True)
    Co, Io = oracle.knn(X, k, metric, True)
    assert I.dtype == torch.int32 and C.shape == (n, k)
    assert torch.equal(I, Io), f"index mismatch rows: {(I != Io).any(1).sum().item()} / {n}"
    assert torch.equal(C, Co), f"distance mismatch: max |d| {(C - Co).abs().max().item()}"


def test_knn_euclidean_and_cross():
    import oracle

    X = gmm(3000, 128, 2.0, seed=1)
    Y = gmm(1777, 128, 2.0, seed=2)
    C, I = _run(X, 20, "euclidean", False, Y)
    Co, Io = oracle.knn(X, 20, "euclidean", False, Y=Y)
    assert torch.equal(I, Io)
    assert torch.equal(C, Co)  # both use a correctly rounded sqrt
    # self-search without exclusion: the point itself is neighbour 0
    C2, I2 = _run(X, 5, "sqeuclidean", False)
    Co2, Io2 = oracle.knn(X, 5, "sqeuclidean", False)
    assert torch.equal(I2, Io2) and torch.equal(C2, Co2)


def test_dense_and_k_ge_n():
    import oracle
    from torchdr_amd.distance import pairwise_distances

    X = gmm(700, 50, 2.0, seed=3)
    _, _, full = oracle.knn(X, 0, "sqeuclidean", False, want_full=True)
    C, I = pairwise_distances(X.cuda(), metric="sqeuclidean", k=700, exclude_diag=True, return_indices=True)
    assert I is None
    expect = full.clone()
    idx = torch.arange(700)
    expect[idx, idx] = expect[idx, idx] + 1e12
    assert torch.equal(C.cpu(), expect)
    C2 = pairwise_distances(X.cuda(), metric="sqeuclidean")
    assert torch.equal(C2.cpu(), full)


def test_knn_property_full_size_rows():
    """Size-independent properties at a larger N: ascending rows, no self, unique indices,
    and a random sample of rows re-checked against the oracle."""
    import oracle

    n, d, k = 100_000, 128, 30
    X = gmm(n, d, 2.0, seed=7)
    C, I = _run(X, k, "sqeuclidean", True)
    assert (C[:, 1:] >= C[:, :-1]).all()
    assert (I != torch.arange(n, dtype=torch.int32)[:, None]).all()
    assert (I.sort(1).values[:, 1:] != I.sort(1).values[:, :-1]).all()
    rows = torch.randperm(n, generator=torch.Generator().manual_seed(0))[:512].sort().values
    for r0 in rows.tolist()[:512:8]:
        Co, Io = oracle.knn(X[r0:r0 + 1], k, "sqeuclidean", True, Y=X, q_offset=r0)
        assert torch.equal(I[r0:r0 + 1], Io) and torch.equal(C[r0:r0 + 1], Co)
