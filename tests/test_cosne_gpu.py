"""COSNE on the HIP path (float64) against the trajectories recorded from the real reference
(tests/golden/cosne.npz) and the CPU oracle (oracle/ref_torch.py: autograd restatement + closed form + RAdam step).

Two recorded runs: ``small_`` (lr = 0.05: the points stay inside the ball, everything is well conditioned -> tight
tolerances) and ``auto_`` (the reference default lr = N/4 = 75: one step throws every point onto the boundary
|z| = 1 - 1e-5, where 1 - |z|^2 ~ 2e-5 and the ball arithmetic amplifies rounding by up to ~1e9 -- the reference's own
autograd and closed forms differ by 5e-10 there, and its optimizer step by 1e-6; tolerances say so)."""

import pytest
import torch

from tests.conftest import gmm
from tests.test_oracle_golden import load

pytestmark = pytest.mark.gpu

KEEP = (0, 1, 2, 5, 19)


def _primed(g, pre, lr):
    """A COSNE instance holding the fixture's affinity graph, ready for ``_training_step`` (no kNN / search run)."""
    import torchdr_amd
    from torchdr_amd.neighbor_embedding.base import build_transposed_graph

    X = g["X"]
    n = X.shape[0]
    m = torchdr_amd.COSNE(perplexity=10, max_iter=20, learning_rate_for_h_loss=0.1, gamma=2, lr=lr, distributed=False)
    dev = torch.device("cuda")
    m.n_samples_in_, m.device_, m.chunk_start_, m.chunk_size_ = n, dev, 0, n
    m.affinity_in_, m.NN_indices_ = g[pre + "P"].to(dev), g[pre + "NN"].to(dev).contiguous()
    m._x_sqnorm = (X ** 2).sum(-1).to(dev)
    m._tgraph = build_transposed_graph(m.affinity_in_, m.NN_indices_, 0, n, 1)
    m.early_exaggeration_coeff_ = m.early_exaggeration_coeff
    m._nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)
    m.lr_ = lr
    m._configure_optimizer()
    m._configure_scheduler()
    return m


@pytest.mark.parametrize("pre,tol_oracle,tol_golden", [("small_", 1e-11, 1e-11), ("auto_", 1e-8, 1e-7)])
def test_gradient_matches_oracle_and_reference(pre, tol_oracle, tol_golden):
    from oracle import ref_torch as R

    g = load("cosne")
    Xn = (g["X"] ** 2).sum(-1)
    m = _primed(g, pre, float(g[pre + "lr"]))
    for it in KEEP:
        Z = g[f"{pre}Zb{it}"]
        m.embedding_ = Z.cuda().contiguous()
        G = m._euclidean_gradient().cpu()
        Go = R.cosne_grad(Z, g[pre + "P"], g[pre + "NN"], Xn, 2.0, 0.1)
        assert float((G - Go).abs().max() / Go.abs().max()) < tol_oracle, f"{pre}{it}"
        # what the reference's autograd left in .grad: the gradient rescaled by 1 / lambda^2 (egrad2rgrad, in place)
        rg = G / R._lambda_x(Z) ** 2
        Rg = g[f"{pre}R{it}"]
        assert float((rg - Rg).abs().max() / Rg.abs().max()) < tol_golden, f"{pre}{it}"


def test_gradient_against_autograd_of_the_restated_loss():
    """Independent of the fixture: random interior points, n_components = 2 .. 8, exaggeration and repulsion weights."""
    from oracle import ref_torch as R
    from torchdr_amd import _lib
    from torchdr_amd.neighbor_embedding.base import build_transposed_graph

    gen = torch.Generator().manual_seed(3)
    for nc, n, k in ((2, 700, 12), (3, 333, 7), (4, 257, 5), (5, 300, 6), (8, 411, 9), (2, 150, 0), (3, 97, 0)):
        Z = R.hyperbolic_init(torch.randn(n, nc, generator=gen, dtype=torch.float64), 0.7)
        if k == 0:       # sparsity=False: the dense affinity as a graph of width n (row i lists 0 .. n - 1, zero diagonal)
            k = n
            NN = torch.arange(n, dtype=torch.int32).unsqueeze(0).expand(n, -1).contiguous()
            P = torch.rand(n, n, generator=gen)
            P.fill_diagonal_(0.0)
        else:
            NN = torch.stack([torch.randperm(n, generator=gen)[:k] for _ in range(n)]).int()
            P = torch.rand(n, k, generator=gen)
        Xn = torch.rand(n, generator=gen) * 5
        Zr = Z.clone().requires_grad_()
        R.cosne_loss(Zr, P, NN, Xn, 1.5, 0.3, exag=4.0, rep=0.7).backward()
        L = _lib.lib()
        Zd, Pd, NNd = Z.cuda(), P.cuda(), NN.cuda()
        tg = build_transposed_graph(Pd, NNd, 0, n, 1)
        nb = int(L.tdr_cosne_workspace_bytes(n, n, nc))
        ws = torch.empty(nb // 8 + 1, dtype=torch.float64, device="cuda")
        rows = torch.empty(n, dtype=torch.float64, device="cuda")
        G = torch.empty((n, nc), dtype=torch.float64, device="cuda")
        st = _lib.stream_ptr()
        _lib.check(L.tdr_cosne_pairs_f64(_lib.ptr(Zd), nc, n, 0, n, 1.5, _lib.ptr(rows), _lib.ptr(ws), nb, st), "pairs")
        S = rows.sum().reshape(1)
        _lib.check(L.tdr_cosne_grad_f64(_lib.ptr(Zd), nc, n, 0, n, _lib.ptr(NNd), _lib.ptr(Pd), k, _lib.ptr(tg[0]),
                                        _lib.ptr(tg[1]), _lib.ptr(tg[2]), _lib.ptr(S), _lib.ptr(Xn.cuda()), 1.5, 0.3, 4.0,
                                        0.7, _lib.ptr(ws), nb, _lib.ptr(G), st), "grad")
        assert float((G.cpu() - Zr.grad).abs().max() / Zr.grad.abs().max()) < 1e-10, f"nc={nc}"


@pytest.mark.parametrize("pre,atol", [("small_", 1e-12), ("auto_", 1e-4)])
def test_radam_step_matches_the_reference_state(pre, atol):
    from oracle import ref_torch as R
    from torchdr_amd.utils.radam import PoincareAdamKernel

    g = load("cosne")
    lr = float(g[pre + "lr"])
    for it in KEEP:
        Z = g[f"{pre}Zb{it}"]
        egrad = g[f"{pre}R{it}"] * R._lambda_x(Z) ** 2            # undo the in-place rescale: the Euclidean gradient
        opt = PoincareAdamKernel(lr=lr)
        opt.exp_avg, opt.exp_avg_sq = g[f"{pre}EAb{it}"].cuda(), g[f"{pre}ESb{it}"].cuda()
        opt.step_count = int(g[f"{pre}stepb{it}"])
        rows = Z.cuda().contiguous()
        rg = opt.step(rows, egrad.cuda())
        assert opt.step_count == int(g[f"{pre}stepa{it}"])
        assert torch.allclose(rows.cpu(), g[f"{pre}Za{it}"], rtol=0, atol=atol), f"{pre}{it}"
        assert torch.allclose(opt.exp_avg_sq.cpu(), g[f"{pre}ESa{it}"], rtol=1e-9, atol=0)
        ea = g[f"{pre}EAa{it}"]
        assert float((opt.exp_avg.cpu() - ea).abs().max() / ea.abs().max()) < (1e-11 if pre == "small_" else 1e-4)
        assert torch.allclose(rg.cpu(), g[f"{pre}R{it}"], rtol=1e-12, atol=0)
        assert float(rows.norm(dim=1).max()) <= 1 - 1e-5 + 1e-12      # projected back into the ball


def test_trajectory_follows_the_reference():
    """20 steps from the reference's initial state (lr = 0.05): gradient kernels + optimizer kernel, end to end."""
    g = load("cosne")
    m = _primed(g, "small_", 0.05)
    m.embedding_ = g["small_Zb0"].cuda().contiguous()
    for it in range(20):
        m.n_iter_.fill_(it)
        m._training_step()
        if it in KEEP:
            assert torch.allclose(m.embedding_.cpu(), g[f"small_Za{it}"], rtol=0, atol=1e-10), f"iteration {it}"
    assert int(m._nan_flag.item()) == 0


def test_estimator_surface():
    import numpy as np
    import torchdr_amd
    from oracle import ref_torch as R

    X = gmm(600, 12, 2.0, seed=9)
    m = torchdr_amd.COSNE(perplexity=15, max_iter=60, lr=0.05, random_state=0, learning_rate_for_h_loss=0.1)
    Z = m.fit_transform(X.cuda())
    assert Z.dtype == torch.float64 and Z.shape == (600, 2) and bool(torch.isfinite(Z).all())
    assert float(Z.norm(dim=1).max()) < 1.0 and int(m.n_iter_) == 59
    # the loss of the restated reference objective went down from the initial state of the same seed
    m0 = torchdr_amd.COSNE(perplexity=15, max_iter=1, lr=1e-12, random_state=0, learning_rate_for_h_loss=0.1)
    Z0 = m0.fit_transform(X.cuda())
    from torchdr_amd.affinity import EntropicAffinity

    P, NN = EntropicAffinity(perplexity=15)(X.cuda(), return_indices=True)
    Xn = (X ** 2).sum(-1)
    l0 = R.cosne_loss(Z0.cpu(), P.cpu(), NN.cpu(), Xn, 2.0, 0.1)
    l1 = R.cosne_loss(Z.cpu(), P.cpu(), NN.cpu(), Xn, 2.0, 0.1)
    assert float(l1) < float(l0)
    # default learning rate (N / 4) as in the reference's own smoke test (tests/test_neighbor_embedding.py:78-94)
    Zd = torchdr_amd.COSNE(perplexity=15, max_iter=30, random_state=0).fit_transform(X.numpy())
    assert isinstance(Zd, np.ndarray) and Zd.dtype == np.float64 and not np.isnan(Zd).any()
    Z3 = torchdr_amd.COSNE(perplexity=15, max_iter=10, n_components=3, lr=0.05).fit_transform(X.cuda())
    assert Z3.shape == (600, 3)
    with pytest.raises(ValueError, match="init pca not supported"):
        torchdr_amd.COSNE(perplexity=15, init="pca").fit_transform(X.cuda())
    Z6 = torchdr_amd.COSNE(perplexity=15, max_iter=10, n_components=6, lr=0.05).fit_transform(X.cuda())
    assert Z6.shape == (600, 6) and bool(torch.isfinite(Z6).all()) and float(Z6.norm(dim=1).max()) < 1.0
    with pytest.raises(NotImplementedError, match="n_components"):
        torchdr_amd.COSNE(perplexity=15, n_components=9).fit_transform(X.cuda())
    # sparsity=False: the dense (N, N) entropic affinity (reference cosne.py:162-171 with NN_indices_ = None)
    md = torchdr_amd.COSNE(perplexity=15, max_iter=40, lr=0.05, random_state=0, sparsity=False)
    Zd2 = md.fit_transform(X.cuda())
    assert Zd2.shape == (600, 2) and bool(torch.isfinite(Zd2).all()) and float(Zd2.norm(dim=1).max()) < 1.0
    Pd, _ = EntropicAffinity(perplexity=15, sparsity=False)(X.cuda(), return_indices=True)
    NNd = torch.arange(600).unsqueeze(0).expand(600, -1)
    md0 = torchdr_amd.COSNE(perplexity=15, max_iter=1, lr=1e-12, random_state=0, sparsity=False)
    Zd0 = md0.fit_transform(X.cuda())
    assert float(R.cosne_loss(Zd2.cpu(), Pd.cpu(), NNd, Xn, 2.0, 0.1)) < float(R.cosne_loss(Zd0.cpu(), Pd.cpu(), NNd, Xn, 2.0, 0.1))
