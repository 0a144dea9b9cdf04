"""K0 (csrc/tdr_prep.hip): the steps either side of the hot path inside fit_transform -- isfinite scan
(utils/validation.py:308), duplicate rows (base.py:132-148, torch.unique(dim=0)), PCA initialisation
(spectral_embedding/pca.py:151-184) -- against their torch formulations."""

import pytest
import torch

from tests.conftest import gmm

pytestmark = pytest.mark.gpu


def test_nonfinite_scan_counts_what_isfinite_rejects():
    from torchdr_amd.utils.validation import count_nonfinite, validate_tensor

    gen = torch.Generator().manual_seed(0)
    for n, d in ((1, 1), (7, 3), (1000, 127), (4097, 128), (33, 1025)):
        X = torch.randn(n, d, generator=gen)
        bad = torch.rand(n, d, generator=gen) < 0.01
        X[bad] = float("inf")
        X[torch.rand(n, d, generator=gen) < 0.005] = float("nan")
        X[torch.rand(n, d, generator=gen) < 0.005] = -float("inf")
        want = int((~torch.isfinite(X)).sum())
        assert count_nonfinite(X.cuda()) == want
        wide = torch.zeros(n, d + 5)
        wide[:, :d] = X
        assert count_nonfinite(wide.cuda()[:, :d]) == want          # row stride > d
        assert count_nonfinite(X.cuda().T) == want                    # non-unit inner stride: counted by torch
    ok = torch.randn(100, 8).cuda()
    assert validate_tensor(ok) is ok
    ok[17, 3] = float("nan")
    with pytest.raises(ValueError, match="infinite"):
        validate_tensor(ok)


def test_duplicate_rows_are_found_like_torch_unique():
    from torchdr_amd.base import unique_rows

    gen = torch.Generator().manual_seed(1)
    X = torch.randn(5000, 24, generator=gen)
    X[100:140] = X[:40]                 # duplicates of earlier rows
    X[4000] = X[4999] = X[2500]         # a triple
    X[7, 3] = 0.0
    X[3000] = X[7]
    X[3000, 3] = -0.0                   # equal by value (-0.0 == 0.0), different bits
    Xu, inv = unique_rows(X.cuda())
    ref_u, ref_inv = torch.unique(X, dim=0, return_inverse=True)
    assert Xu.shape[0] == ref_u.shape[0] == 5000 - 39 - 2 - 1   # row 107 stopped duplicating row 7 when X[7, 3] changed
    assert bool((Xu[inv].cpu() == X).all())   # re-expansion gives back every row (by value: -0.0 == 0.0)
    # same partition into groups of equal rows as torch.unique, and first occurrences keep their order
    same_ref = ref_inv[:, None] == ref_inv[None, :100]
    same_got = inv.cpu()[:, None] == inv.cpu()[None, :100]
    assert torch.equal(same_ref, same_got)
    first = torch.tensor([int((inv.cpu() == g).nonzero()[0]) for g in range(0, Xu.shape[0], 97)])
    assert bool((first[1:] > first[:-1]).all())
    # no duplicates: the input itself comes back
    Y = torch.randn(3000, 16, generator=gen).cuda()
    Yu, none = unique_rows(Y)
    assert none is None and Yu.data_ptr() == Y.data_ptr()


def test_pca_initialisation_matches_the_torch_formulation():
    from torchdr_amd import _lib
    from torchdr_amd.affinity_matcher import pca_scores

    for n, d in ((5000, 50), (20000, 128), (3000, 200), (1000, 256), (500, 7)):
        X = gmm(n, d, 2.0, seed=n).cuda()
        L = _lib.lib()
        mean = torch.empty(d, device="cuda")
        G = torch.empty((d, d), dtype=torch.float64, device="cuda")
        wsf = int(L.tdr_pca_gram_workspace_floats(n, d))
        ws = torch.empty(wsf, device="cuda")
        _lib.check(L.tdr_pca_gram_f32(_lib.ptr(X), n, d, X.stride(0), _lib.ptr(mean), _lib.ptr(G), _lib.ptr(ws), wsf,
                                      _lib.stream_ptr()), "gram")
        Xd = X.double()
        mu = Xd.mean(0)
        Gd = (Xd - mu).T @ (Xd - mu)
        assert torch.allclose(mean.double(), mu, rtol=1e-5, atol=1e-6)
        assert torch.allclose(G, Gd, rtol=1e-4, atol=1e-4 * float(Gd.abs().max()))
        assert torch.equal(G, G.T.contiguous()) or torch.allclose(G, G.T, rtol=1e-6)
        # scores: same subspace and signs as the SVD route of the reference (pca.py:169-178, svd_flip u-based)
        E = pca_scores(X, 2)
        U, S, Vt = torch.linalg.svd(Xd - mu, full_matrices=False)
        ref = U[:, :2] * S[:2]
        idx = ref.abs().argmax(0)
        sg = torch.sign(ref[idx, torch.arange(2, device="cuda")])
        ref = ref * sg
        assert torch.allclose(E.double(), ref, rtol=1e-3, atol=2e-4 * float(ref.abs().max())), (n, d)


@pytest.mark.parametrize("d", [1, 2, 3, 7, 50, 128, 129, 256])
def test_jacobi_eigh_vs_lapack(d):
    """tdr_eigh_jacobi_f64 (the D x D eigenproblem of the PCA initialisation) against LAPACK in float64 on the host:
    eigenvalues descending to 1e-12 of the largest, eigenvector residual |G V - V diag(lambda)| and orthogonality at the
    same level; a rank-deficient Gram matrix (more features than samples) included."""
    from torchdr_amd import _lib

    gen = torch.Generator().manual_seed(d)
    n = 40 if d in (50, 129) else 2000          # 40 < d: rank-deficient
    X = torch.randn(n, d, generator=gen, dtype=torch.float64) * (torch.rand(d, generator=gen, dtype=torch.float64) * 3 + 0.1)
    X = X - X.mean(0)
    G = (X.T @ X).contiguous()
    Gd = G.cuda()
    evals = torch.empty(d, dtype=torch.float64, device="cuda")
    evecs = torch.empty((d, d), dtype=torch.float64, device="cuda")
    ws = torch.empty(2 * d * d, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().tdr_eigh_jacobi_f64(_lib.ptr(Gd), d, _lib.ptr(evals), _lib.ptr(evecs), _lib.ptr(ws), _lib.stream_ptr()),
               "tdr_eigh_jacobi_f64")
    ev, V = evals.cpu(), evecs.cpu()
    ref = torch.linalg.eigvalsh(G).flip(0).clamp_min(0)
    scale = float(ref[0])
    assert bool((ev[:-1] >= ev[1:]).all())
    assert float((ev - ref).abs().max()) < 1e-12 * scale
    assert float((G @ V - V * ev[None, :]).abs().max()) < 1e-11 * scale
    assert float((V.T @ V - torch.eye(d, dtype=torch.float64)).abs().max()) < 1e-11


@pytest.mark.parametrize("d", [1, 2, 3, 7, 50, 64, 128, 129, 200, 256])
def test_top_eigh_vs_lapack(d):
    """tdr_eigh_top_f64 (the leading pairs of the PCA initialisation's eigenproblem: Householder + Sturm multisection + inverse
    iteration) against LAPACK in float64 on the host: eigenvalues to 1e-12 of the largest, residual |G V - V diag(lambda)| and
    orthonormality at the same level, for every nc the kernel serves; rank-deficient Gram matrices included."""
    from torchdr_amd import _lib

    gen = torch.Generator().manual_seed(100 + d)
    n = 40 if d in (50, 129) else 2000          # 40 < d: rank-deficient
    X = torch.randn(n, d, generator=gen, dtype=torch.float64) * (torch.rand(d, generator=gen, dtype=torch.float64) * 3 + 0.1)
    X = X - X.mean(0)
    G = (X.T @ X).contiguous()
    ref = torch.linalg.eigvalsh(G).flip(0)
    scale = float(ref[0])
    Gd = G.cuda()
    ws = torch.empty(d * d, dtype=torch.float64, device="cuda")
    for nc in range(1, min(d, 4) + 1):
        evals = torch.empty(nc, dtype=torch.float64, device="cuda")
        evecs = torch.empty((d, nc), dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib().tdr_eigh_top_f64(_lib.ptr(Gd), d, nc, _lib.ptr(evals), _lib.ptr(evecs), _lib.ptr(ws), _lib.stream_ptr()),
                   "tdr_eigh_top_f64")
        ev, V = evals.cpu(), evecs.cpu()
        assert float((ev - ref[:nc]).abs().max()) < 1e-12 * scale, (d, nc)
        assert float((G @ V - V * ev[None, :]).abs().max()) < 1e-11 * scale, (d, nc)
        assert float((V.T @ V - torch.eye(nc, dtype=torch.float64)).abs().max()) < 1e-11, (d, nc)
    assert torch.equal(Gd.cpu(), G)       # the input is left alone


def test_top_eigh_degenerate_spectra():
    """Equal leading eigenvalues (any orthonormal basis of the eigenspace is an answer: residual + orthonormality), an already
    diagonal matrix (every Householder step is the identity), the zero matrix, and the bad-argument returns."""
    from torchdr_amd import _lib

    L = _lib.lib()
    gen = torch.Generator().manual_seed(5)
    Q, _ = torch.linalg.qr(torch.randn(40, 40, generator=gen, dtype=torch.float64))
    cases = [Q @ torch.diag(torch.cat([torch.full((3,), 7.0), torch.ones(37)]).double()) @ Q.T,
             torch.diag(torch.tensor([5.0, 5.0, 5.0, 1.0, 1.0], dtype=torch.float64)),
             torch.ones(5, 5, dtype=torch.float64), torch.zeros(4, 4, dtype=torch.float64)]
    for G in cases:
        G = ((G + G.T) / 2).contiguous()
        d = G.shape[0]
        ref = torch.linalg.eigvalsh(G).flip(0)
        evals = torch.empty(4, dtype=torch.float64, device="cuda")
        evecs = torch.empty((d, 4), dtype=torch.float64, device="cuda")
        ws = torch.empty(d * d, dtype=torch.float64, device="cuda")
        _lib.check(L.tdr_eigh_top_f64(_lib.ptr(G.cuda()), d, 4, _lib.ptr(evals), _lib.ptr(evecs), _lib.ptr(ws), _lib.stream_ptr()), "top")
        ev, V = evals.cpu(), evecs.cpu()
        scale = max(float(ref[0].abs()), 1.0)
        assert float((ev - ref[:4]).abs().max()) < 1e-12 * scale
        assert float((G @ V - V * ev[None, :]).abs().max()) < 1e-11 * scale
        assert float((V.T @ V - torch.eye(4, dtype=torch.float64)).abs().max()) < 1e-11
    g = torch.eye(3, dtype=torch.float64, device="cuda")
    out = torch.empty(16, dtype=torch.float64, device="cuda")
    assert L.tdr_eigh_top_f64(_lib.ptr(g), 3, 4, _lib.ptr(out), _lib.ptr(out), _lib.ptr(out), None) != 0      # nc > d
    assert L.tdr_eigh_top_f64(_lib.ptr(g), 3, 0, _lib.ptr(out), _lib.ptr(out), _lib.ptr(out), None) != 0
    assert L.tdr_eigh_top_f64(_lib.ptr(g), 257, 2, _lib.ptr(out), _lib.ptr(out), _lib.ptr(out), None) != 0


def test_pca_scores_same_with_every_eigensolver():
    from torchdr_amd import affinity_matcher as am

    for n, d, nc in ((20000, 64, 2), (5000, 256, 3), (3000, 100, 4)):
        X = gmm(n, d, 2.0, seed=9).cuda()
        old = am.PCA_EIGH
        got = {}
        try:
            for mode in ("top", "jacobi", "library"):
                am.PCA_EIGH = mode
                got[mode] = am.pca_scores(X, nc)
                assert am.pca_scores.deterministic == (mode != "library")
        finally:
            am.PCA_EIGH = old
        b = got["library"]
        for mode in ("top", "jacobi"):
            assert torch.allclose(got[mode], b, rtol=1e-4, atol=1e-4 * float(b.abs().max())), (mode, n, d)
        # same bits from call to call (what lets a row-sharded fit skip the reference's broadcast of the initial embedding)
        am.PCA_EIGH = "top"
        try:
            assert torch.equal(am.pca_scores(X, nc), got["top"])
        finally:
            am.PCA_EIGH = old
