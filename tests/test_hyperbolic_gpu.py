"""metric='sqhyperbolic' (reference distance/torch.py:101-107, distance/base.py:372-377, 392-398) on the HIP path:
library GEMM + HIP epilogue / running top-k.  fp32 like the reference; `acoshf` differs from ATen's vectorised one by
a few ulp, so values agree to 1e-5 relative and neighbour lists can swap only where two distances nearly tie."""

import pytest
import torch

from tests.test_oracle_golden import boundary_safe_rows, load

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def test_knn_dense_and_gathered_forms():
    from torchdr_amd.distance import pairwise_distances, pairwise_distances_indexed

    g = load("hyperbolic")
    X, Y = g["X"].cuda(), g["Y"].cuda()
    C, I = pairwise_distances(X, metric="sqhyperbolic", k=9, exclude_diag=True, return_indices=True)
    assert I.dtype == torch.int32 and torch.allclose(C.cpu(), g["knn_C"], rtol=RTOL, atol=1e-7)
    Cw = g["knn_Cw"]
    clear = (Cw[:, 1:] - Cw[:, :-1]).min(1).values > 1e-4 * Cw[:, -1]          # rows without near-ties in the top 17
    assert clear.float().mean() > 0.8 and torch.equal(I.cpu()[clear].long(), g["knn_I"][clear].long())
    safe = boundary_safe_rows(Cw, 9)
    assert float((I.cpu().long().sort(1).values != g["knn_I"].long().sort(1).values).any(1)[safe].float().mean()) < 0.01
    Cx, Ix = pairwise_distances(X, Y, metric="sqhyperbolic", k=6, return_indices=True)
    assert torch.allclose(Cx.cpu(), g["cross_C"], rtol=RTOL, atol=1e-7)
    assert float((Ix.cpu().long() != g["cross_I"].long()).any(1).float().mean()) < 0.02
    D = pairwise_distances(X, Y, metric="sqhyperbolic")
    assert torch.allclose(D.cpu(), g["cross_dense"], rtol=RTOL, atol=1e-7)
    De = pairwise_distances(X[:120], metric="sqhyperbolic", exclude_diag=True)
    assert torch.allclose(De.cpu(), g["dense_excl"], rtol=RTOL, atol=1e-5) and float(De.diagonal().min()) > 1e11
    q, keys = g["q"].cuda(), g["keys"].cuda()
    G = pairwise_distances_indexed(X, query_indices=q, key_indices=keys, metric="sqhyperbolic")
    assert torch.allclose(G.cpu(), g["indexed"], rtol=1e-4, atol=1e-6)       # direct difference: own rounding of s
    B = pairwise_distances_indexed(X, query_indices=q, key_indices=torch.arange(5, 90).cuda(), metric="sqhyperbolic")
    assert torch.allclose(B.cpu(), g["block"], rtol=1e-4, atol=1e-6)


def test_against_the_torch_restatement_at_other_shapes():
    from oracle import ref_torch as R
    from torchdr_amd.distance import pairwise_distances

    gen = torch.Generator().manual_seed(0)
    for n, d, k in ((3000, 2, 15), (1200, 16, 30), (700, 300, 10)):
        v = torch.randn(n, d, generator=gen)
        X = (v / v.norm(dim=1, keepdim=True) * 0.95 * torch.rand(n, 1, generator=gen) ** (1.0 / d)).contiguous()
        C, I = pairwise_distances(X.cuda(), metric="sqhyperbolic", k=k, exclude_diag=True, return_indices=True)
        Co, Io = R.knn_chunked(X, k, "sqhyperbolic", True)
        assert torch.allclose(C.cpu(), Co, rtol=1e-4, atol=1e-6)
        assert float((I.cpu() != Io).any(1).float().mean()) < 0.03
