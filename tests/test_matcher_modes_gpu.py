"""SinkhornAffinity on input points (Gaussian base kernel, the class default) against the real
reference, and the AffinityMatcher modes beside the closed-form kernels: ``affinity_in="precomputed"`` and the
autograd-loss mode (subclasses that define a loss instead of gradients, reference affinity_matcher.py:418-459)."""

import pytest
import torch

from tests.conftest import gmm
from tests.test_oracle_golden import load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,kw", [("gauss", dict(eps=5.0, base_kernel="gaussian")),
                                     ("gauss_nozd", dict(eps=20.0, base_kernel="gaussian", zero_diag=False)),
                                     ("student", dict(eps=2.0, base_kernel="student"))])
def test_sinkhorn_on_input_points_vs_reference(name, kw):
    from torchdr_amd.affinity import SinkhornAffinity

    g = load("sinkhorn")
    X = g["X"].cuda()
    aff = SinkhornAffinity(tol=1e-5, max_iter=300, **kw)
    logP = aff(X, log=True)
    assert int(aff.n_iter_) == int(g[f"{name}_n_iter"])
    assert torch.allclose(aff.dual_.cpu(), g[f"{name}_dual"], rtol=1e-4, atol=1e-4)
    ref = g[f"{name}_logP_rows"]
    got = logP[:6].cpu()
    live = ref > -1e6      # the excluded diagonal sits at -1e12 / eps
    assert torch.allclose(got[live], ref[live], rtol=1e-4, atol=1e-3)
    # doubly stochastic: rows of N * P sum to 1 (reference test_affinity.py:294)
    assert torch.allclose(logP.exp().sum(1).cpu() * X.shape[0], torch.ones(X.shape[0]), atol=1e-3)


def test_sinkhorn_warm_start_three_iterations():
    from torchdr_amd.affinity import SinkhornAffinity

    g = load("sinkhorn")
    aff = SinkhornAffinity(eps=5.0, tol=1e-5, max_iter=3)
    aff.fit_dual(g["X"].cuda(), init_dual=g["warm_init"].cuda())
    assert torch.allclose(aff.dual_.cpu(), g["warm_dual"], rtol=1e-5, atol=1e-5)


def test_precomputed_affinity_and_autograd_loss_mode():
    """A subclass that defines ``_compute_loss`` with torch ops on the device tensors (no closed-form gradient) on a
    precomputed (n, n) affinity: three SGD steps must equal the same loop written by hand with autograd."""
    from torchdr_amd.affinity_matcher import AffinityMatcher

    n = 200
    gen = torch.Generator().manual_seed(0)
    A = torch.rand(n, n, generator=gen)
    A = ((A + A.T) / 2).cuda()
    Z0 = torch.randn(n, 2, generator=gen).cuda()

    def loss_of(Z, P):
        D = ((Z[:, None, :] - Z[None, :, :]) ** 2).sum(-1)
        return ((1.0 / (1.0 + D) - P) ** 2).sum()

    class Stress(AffinityMatcher):
        def _compute_loss(self):
            return loss_of(self.embedding_, self.affinity_in_)

    for opt, okw in (("SGD", None), ("Adam", None)):
        m = Stress(affinity_in="precomputed", optimizer=opt, optimizer_kwargs=okw, lr=0.01, max_iter=3, init=Z0.clone(),
                   init_scaling=1.0, min_grad_norm=0.0)
        Z = m.fit_transform(A)
        # by hand: same init scaling rule (A.5: Z0 / std(Z0[:, 0])), same optimizer
        Zh = (Z0 / Z0[:, 0].std()).clone().requires_grad_(True)
        o = getattr(torch.optim, opt)([Zh], lr=0.01)
        for _ in range(3):
            o.zero_grad()
            loss_of(Zh, A).backward()
            o.step()
        assert torch.allclose(Z, Zh.detach(), rtol=1e-5, atol=1e-6), opt
    with pytest.raises(ValueError, match="precomputed"):
        Stress(affinity_in="precomputed", max_iter=1).fit_transform(torch.rand(10, 4).cuda())
    with pytest.raises(ValueError, match="negative"):
        Stress(affinity_in="precomputed", max_iter=1).fit_transform(-torch.rand(5, 5).cuda())
    with pytest.raises(ValueError, match="not supported"):
        Stress(affinity_in="precomputed", loss_fn="nope")


def test_neighbor_embedding_loss_hooks_drive_the_autograd_mode():
    """``_compute_attractive_loss`` / ``_compute_repulsive_loss`` (reference neighbor_embedding/base.py:207-231) of a
    user subclass: LargeVis' losses written with torch ops must give the gradient of the closed-form HIP kernel."""
    import torchdr_amd
    from torchdr_amd.neighbor_embedding.base import NegativeSamplingNeighborEmbedding
    from torchdr_amd.affinity import EntropicAffinity

    X = gmm(1500, 16, 3.0, seed=2).cuda()
    n = X.shape[0]
    seen = {}

    class LossLargeVis(NegativeSamplingNeighborEmbedding):
        def _compute_attractive_loss(self):
            Z, NN, P = self.embedding_, self.NN_indices_.long(), self.affinity_in_
            d = ((Z[:, None, :] - Z[NN]) ** 2).sum(-1)
            return (P * (2.0 + d).log()).sum()          # -sum P log Q, Q = 1 / (2 + d)

        def _compute_repulsive_loss(self):
            Z, neg = self.embedding_, self.neg_indices_.long()
            d = ((Z[:, None, :] - Z[neg]) ** 2).sum(-1)
            q = 1.0 / (2.0 + d)
            return -(1.0 - q).log().sum() / n

        def on_training_step_start(self):
            super().on_training_step_start()
            gen = torch.Generator(device="cuda").manual_seed(int(self.n_iter_))
            r = torch.randint(0, n - 1, (n, 5), device="cuda", generator=gen)
            self.neg_indices_ = r + (r >= torch.arange(n, device="cuda")[:, None]).long()

        def _sgd_kernel(self, Z, grad, chunk=False):
            seen[int(self.n_iter_)] = (self.embedding_.detach().clone(), grad.clone(), self.neg_indices_.clone())
            super()._sgd_kernel(Z, grad, chunk)

    aff = EntropicAffinity(perplexity=5, sparsity=True)
    m = LossLargeVis(affinity_in=aff, n_negatives=5, lr=1.0, optimizer="SGD", optimizer_kwargs=None, max_iter=2, random_state=0)
    Z = m.fit_transform(X)
    assert torch.isfinite(Z).all() and set(seen) == {0, 1}
    # the closed-form kernel on the same state and negatives
    from torchdr_amd import _lib
    from torchdr_amd.neighbor_embedding.base import build_transposed_graph

    lv = torchdr_amd.LargeVis(perplexity=5, random_state=0)
    P, NN = lv.affinity_in(X, return_indices=True)
    tg = build_transposed_graph(P, NN.to(torch.int32), 0, n, 1)
    Zs, g_auto, neg = seen[1]
    grad = torch.zeros((n, 2), device="cuda")
    _lib.check(_lib.lib().tdr_ne_grad_f32(_lib.ptr(Zs.contiguous()), 2, n, 0, n, _lib.ptr(NN.to(torch.int32).contiguous()),
                                          _lib.ptr(P.contiguous()), P.shape[1], _lib.ptr(tg[0]), _lib.ptr(tg[1]), _lib.ptr(tg[2]), 0, 1.0,
                                          2.0 / n, 5, _lib.ptr(neg.contiguous()), 0, 1, _lib.ptr(grad), _lib.stream_ptr()), "ne_grad")
    assert torch.allclose(grad, g_auto, rtol=1e-4, atol=1e-6 * float(g_auto.abs().max()))
