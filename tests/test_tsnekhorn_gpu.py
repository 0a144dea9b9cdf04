"""K7 / K8 parity (GPU): matrix-free SEA, Sinkhorn and the TSNEkhorn force vs golden vectors of the reference."""

import numpy as np
import pytest
import torch

from tests.conftest import gmm, grade32, grade64
from tests.test_oracle_golden import load

pytestmark = pytest.mark.gpu
BUDGET = 1e-5   # of max |g|, against the float64 evaluation of the reference's loss (tests/golden/grad64.npz)


def test_sea_matrix_free_vs_reference():
    from torchdr_amd.affinity import SymmetricEntropicAffinity

    g = load("tsnekhorn")
    X = g["X"].cuda()
    n = X.shape[0]
    sea = SymmetricEntropicAffinity(perplexity=10, lr=1e-1, max_iter=30, tol=1e-3, zero_diag=False)
    logP = sea(X, log=True)
    assert int(sea.n_iter_) == int(g["sea_n_iter"])
    assert torch.allclose(sea.eps_.cpu(), g["sea_eps"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(sea.mu_.cpu(), g["sea_mu"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(logP.cpu(), g["sea_logP"], rtol=1e-4, atol=2e-3)   # log of entries down to e^-80: the reference's float32 log P is itself 1.3e-2 from its float64 run
    assert torch.allclose(logP.exp().cpu(), g["sea_logP"].exp(), rtol=2e-3, atol=1e-9)
    assert torch.allclose(logP, logP.T, atol=1e-5)  # symmetry (reference test_affinity.py:260)
    # against the reference run in FLOAT64 on the same points: 29 Adam steps on the duals amplify float32 rounding -- the
    # reference's own float32 run is 9.4e-6 off in eps and 2.1e-5 of max P off in P (make_golden.py grad64 prints both)
    g64 = load("grad64")
    grade64("tsnekhorn/sea_eps", sea.eps_, g64["tsnekhorn/sea64_eps"], 2e-5)          # measured 9.4e-6 = the reference's own float32 run
    grade64("tsnekhorn/sea_P", logP.exp(), g64["tsnekhorn/sea64_logP"].exp(), 3e-5)   # measured 2.09e-5 = the reference's own float32 run
    grade32("tsnekhorn/sea_P/vs_reference_float32", logP.exp(), g["sea_logP"].exp(), BUDGET)
    seaz = SymmetricEntropicAffinity(perplexity=10, lr=1e-1, max_iter=8, tol=1e-3, zero_diag=True)
    lz = seaz(X, log=True).cpu()
    off = ~torch.eye(n, dtype=torch.bool)
    assert torch.allclose(lz[off], g["sea_zd_logP"][off], rtol=1e-4, atol=2e-3)


def test_sea_rowstats_vs_oracle_medium():
    """Row statistics at a size with many tiles / several workgroups, against a dense fp64 evaluation."""
    import oracle
    from torchdr_amd.affinity.entropic import sea_rowstats
    from torchdr_amd.distance import PackedPoints

    X = gmm(3000, 64, 2.0, seed=8)
    _, _, C = oracle.knn(X, 0, "sqeuclidean", False, want_full=True)
    gen = torch.Generator().manual_seed(1)
    mu = torch.rand(3000, generator=gen) * 2 - 1
    e = torch.rand(3000, generator=gen) * 20 + 40
    lp = (mu[:, None] + mu[None, :] - 2 * C.double()) / (e[:, None] + e[None, :]).double()
    S_ref = lp.exp().sum(1)
    H_ref = -(lp.exp() * (lp - 1)).sum(1)
    S, H = sea_rowstats(PackedPoints(X.cuda()), mu.cuda(), e.cuda(), False)
    assert torch.allclose(S.cpu().double(), S_ref, rtol=1e-5)
    assert torch.allclose(H.cpu().double(), H_ref, rtol=1e-5, atol=1e-4)


def test_split_pair_scan_equals_the_unsplit_scan():
    """With the optional workspace the database is scanned in segments by separate workgroups and the per-segment statistics
    are folded in order: same numbers up to the association of the fp32 sums (row statistics, log-sum-exp, forces)."""
    from torchdr_amd import _lib
    from torchdr_amd.affinity.entropic import pair_scan_workspace
    from torchdr_amd.distance import PackedPoints

    n = 20000
    X = gmm(n, 64, 2.0, seed=3).cuda()
    packed = PackedPoints(X)
    gen = torch.Generator().manual_seed(1)
    mu = (torch.rand(n, generator=gen) * 2 - 1).cuda()
    e = (torch.rand(n, generator=gen) * 20 + 40).cuda()
    L = _lib.lib()
    assert int(L.tdr_pair_scan_workspace_bytes(3000, 4)) == 0 and int(L.tdr_pair_scan_workspace_bytes(n, 4)) > 0
    side = torch.stack([mu, e], dim=1).contiguous()
    out = {}
    for split in (False, True):
        ws, nb, keep = pair_scan_workspace(n, 4, X.device) if split else (None, 0, None)
        S, H, En = (torch.empty(n, device="cuda") for _ in range(3))
        _lib.check(L.tdr_sea_rowstats3_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(side), 1, 1e12, _lib.ptr(S), _lib.ptr(H),
                                           _lib.ptr(En), ws, nb, _lib.stream_ptr()), "rowstats3")
        f = (mu * 0.3).contiguous()
        lse = torch.empty(n, device="cuda")
        ws2, nb2, keep2 = pair_scan_workspace(n, 2, X.device) if split else (None, 0, None)
        _lib.check(L.tdr_sinkhorn_lse_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(f), 0.05, 0, 1, 1e12, _lib.ptr(lse), ws2, nb2,
                                          _lib.stream_ptr()), "lse")
        Z = torch.randn(n, 3, generator=torch.Generator().manual_seed(2)).cuda() * 3
        side3 = torch.cat([mu[:, None], e[:, None], Z, (0.1 * mu).exp()[:, None]], dim=1).contiguous()
        grad = torch.empty((n, 3), device="cuda")
        ws3, nb3, keep3 = pair_scan_workspace(n, 3, X.device) if split else (None, 0, None)
        _lib.check(L.tdr_khorn_grad_nc_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(side3), 3, float(np.log(n)), _lib.ptr(grad),
                                           ws3, nb3, _lib.stream_ptr()), "khorn")
        torch.cuda.synchronize()
        out[split] = (S, H, En, lse, grad)
    for name_, a_, b_ in zip(("rowsum", "entropy", "energy", "lse", "force"), out[False], out[True]):
        if name_ == "force":     # signed terms that cancel: the tolerance of the force tests below
            assert torch.allclose(a_, b_, rtol=2e-3, atol=2e-5 * float(a_.abs().max()))
            continue
        rel = float(((a_ - b_).abs() / (a_.abs() + 1e-6 * float(a_.abs().max()))).max())
        assert rel < 5e-5, (name_, rel)      # fp32 sums of 20 000 positive terms in two associations
    # a buffer that is too small is ignored (unsplit scan, bit-equal to no buffer)
    small = torch.empty(1024, dtype=torch.uint8, device="cuda")
    S2, H2 = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    _lib.check(L.tdr_sea_rowstats_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(side), 1, 1e12, _lib.ptr(S2), _lib.ptr(H2),
                                      _lib.ptr(small), 1024, _lib.stream_ptr()), "rowstats")
    assert torch.equal(S2, out[False][0]) and torch.equal(H2, out[False][1])


@pytest.mark.parametrize("nc", [2, 3, 8])
def test_split_passes_on_the_embedding_equal_the_unsplit_ones(nc):
    """Sinkhorn update and adjoint mat-vec on the embedding with the columns spread over several workgroups per row block
    (workspace given) against the one-workgroup-per-row-block launches, and against a float64 evaluation on sampled rows."""
    from torchdr_amd import _lib

    L = _lib.lib()
    n = 20_011
    gen = torch.Generator().manual_seed(3)
    Z = (torch.randn(n, nc, generator=gen) * 3).cuda().contiguous()
    f = (torch.randn(n, generator=gen) * 0.3).cuda()
    v = torch.randn(n, generator=gen).cuda()
    fmax = float(f.max())
    Ef = (f - fmax).exp()
    nb = int(L.tdr_student_workspace_bytes(n))
    assert nb > 0 and int(L.tdr_student_workspace_bytes(1500)) == 0
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    res = {}
    for split in (False, True):
        fn, r2, out = torch.empty(n, device="cuda"), torch.zeros(1, device="cuda"), torch.empty(n, device="cuda")
        w, b_ = (_lib.ptr(ws), nb) if split else (None, 0)
        _lib.check(L.tdr_sinkhorn_pass_f32(_lib.ptr(Z), nc, _lib.ptr(f), _lib.ptr(Ef), fmax, n, 1, 1e12, _lib.ptr(fn), _lib.ptr(r2), w, b_,
                                           _lib.stream_ptr()), "pass")
        _lib.check(L.tdr_student_matvec_f32(_lib.ptr(Z), nc, _lib.ptr(v), n, 1, 1e12, _lib.ptr(out), w, b_, _lib.stream_ptr()), "matvec")
        torch.cuda.synchronize()
        res[split] = (fn, r2, out)
    assert torch.allclose(res[False][0], res[True][0], rtol=1e-5, atol=1e-5), float((res[False][0] - res[True][0]).abs().max())
    assert abs(float(res[False][1]) - float(res[True][1])) < 1e-4 * float(res[False][1])
    scale = float(res[False][2].abs().max())
    grade32(f"student_matvec_split_vs_unsplit/nc={nc}", res[True][2], res[False][2], BUDGET)   # float32 vs float32: 20 000 signed terms, two associations
    sub = torch.arange(0, n, 97)
    Zd = Z.double().cpu()
    W = 1 / (1 + ((Zd[sub, None, :] - Zd[None, :, :]) ** 2).sum(-1))
    W[torch.arange(len(sub)), sub] = 0      # zero_diag: the diagonal term is weighted 1 / (1 + 1e12)
    ref = W @ v.double().cpu()
    grade64(f"student_matvec_vs_float64/nc={nc}", res[True][2].cpu()[sub], ref, BUDGET)
    red = -(fmax + (W @ Ef.double().cpu()).log())
    assert torch.allclose(res[True][0].cpu().double()[sub], 0.5 * (f.double().cpu()[sub] + red), rtol=1e-5, atol=1e-5)


def test_sinkhorn_student_vs_reference():
    from torchdr_amd.affinity import SinkhornAffinity

    g = load("tsnekhorn")
    sk = SinkhornAffinity(base_kernel="student", max_iter=5)
    logQ = sk(g["sk_Z"].cuda(), log=True, init_dual=g["sk_init"].cuda())
    assert torch.allclose(sk.dual_.cpu(), g["sk_dual"], rtol=1e-5, atol=2e-6)
    assert torch.allclose(logQ.cpu(), g["sk_logQ"], rtol=1e-5, atol=1e-5)
    # run to convergence: doubly stochastic (rows of Q sum to 1/n) -- reference test_affinity.py:294
    sk2 = SinkhornAffinity(base_kernel="student", max_iter=1000, tol=1e-5)
    Q = sk2(g["sk_Z"].cuda())
    assert torch.allclose(Q.sum(1).cpu(), torch.full((256,), 1 / 256), rtol=1e-3)
    # with_grad=True: same numbers; an input that requires grad is refused (no autograd graph on the HIP path)
    skg = SinkhornAffinity(base_kernel="student", max_iter=5, with_grad=True)
    assert torch.equal(skg(g["sk_Z"].cuda(), log=True, init_dual=g["sk_init"].cuda()), logQ)
    with pytest.raises(NotImplementedError):
        skg(g["sk_Z"].cuda().requires_grad_(True))
    # the Gaussian base kernel (class default) runs the matrix-free pair scan: tests/test_matcher_modes_gpu.py


def test_tsnekhorn_gradient_and_steps_vs_reference_autograd():
    import torchdr_amd
    from torchdr_amd import _lib
    from torchdr_amd.affinity.entropic import sinkhorn_student_dual
    from torchdr_amd.distance import PackedPoints

    g = load("tsnekhorn")
    X = g["X"].cuda()
    n = X.shape[0]
    sea = torchdr_amd.SymmetricEntropicAffinity(perplexity=10, lr=1e-1, max_iter=30, tol=1e-3, zero_diag=False)
    packed = sea.fit_duals(X)
    mu, e = sea.dual_side()
    init = None
    for t in range(2):
        Z = g[f"tk_Z_{t}"].cuda().contiguous()
        dual, _ = sinkhorn_student_dual(Z, init, 5, 1e-5, True)
        init = dual
        assert torch.allclose(dual.cpu(), g[f"tk_dual_{t}"], rtol=1e-5, atol=2e-6)
        side = torch.stack([mu, e, Z[:, 0], Z[:, 1], dual.exp()], dim=1).contiguous()
        grad = torch.empty((n, 2), device="cuda")
        _lib.check(_lib.lib().tdr_khorn_grad_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(side), float(np.log(n)),
                                                 _lib.ptr(grad), _lib.stream_ptr()), "khorn")
        ref = g[f"tk_grad_{t}"]
        grade64(f"tsnekhorn/tk_{t}", grad, load("grad64")[f"tsnekhorn/tk_grad64_{t}"], BUDGET)
        grade32(f"tsnekhorn/tk_{t}/vs_reference_float32", grad, ref, BUDGET)


def _khorn_side(X):
    import torchdr_amd

    sea = torchdr_amd.SymmetricEntropicAffinity(perplexity=10, lr=1e-1, max_iter=30, tol=1e-3, zero_diag=False)
    packed = sea.fit_duals(X)
    mu, e = sea.dual_side()
    return packed, mu, e


@pytest.mark.parametrize("name", ["u2", "u3", "u5"])
def test_tsnekhorn_unrolled_gradient_vs_reference_autograd(name):
    """TSNEkhorn(unrolling=True) (tsnekhorn.py:134, 224-227): the reference back-propagates through the 5 Sinkhorn updates;
    here the recorded passes + 5 adjoint mat-vecs + the bilinear force kernel, against the reference's autograd gradient of
    two warm-started steps (2 / 3 components exact instances, 5 zero-padded to 8)."""
    from torchdr_amd import _lib
    from torchdr_amd.affinity.entropic import pad_embedding, sea_rowstats, sinkhorn_student_adjoint, sinkhorn_student_dual

    g = load("tsnekhorn_unrolled")
    X = g["X"].cuda()
    n = X.shape[0]
    packed, mu, e = _khorn_side(X)
    S, _ = sea_rowstats(packed, mu, e, False)
    assert torch.allclose((S / n).cpu(), g["log_P"].exp().sum(1), rtol=2e-4)
    init = None
    for t in range(2):
        Z = g[f"{name}_Z_{t}"].cuda()
        nc = Z.shape[1]
        Zp = pad_embedding(Z)
        rec = []
        dual, k = sinkhorn_student_dual(Zp, init, 5, 1e-5, True, record=rec)
        init = dual
        assert k == 4 and len(rec) == 5
        assert torch.allclose(dual.cpu(), g[f"{name}_dual_{t}"], rtol=1e-5, atol=2e-6)
        A, B = sinkhorn_student_adjoint(Zp, rec, -(2.0 / n) * S, True)
        side = torch.cat([mu[:, None], e[:, None], Zp, 0.25 * A, B], dim=1).contiguous()
        grad = torch.empty((n, Zp.shape[1]), device="cuda")
        _lib.check(_lib.lib().tdr_khorn_grad_unrolled_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(side), Zp.shape[1],
                                                          float(np.log(n)), _lib.ptr(grad), None, 0, _lib.stream_ptr()), "khorn unrolled")
        ref = g[f"{name}_grad_{t}"]
        grade64(f"tsnekhorn_unrolled/{name}_{t}", grad[:, :nc], load("grad64")[f"tsnekhorn_unrolled/{name}_grad64_{t}"], BUDGET)
        grade32(f"tsnekhorn_unrolled/{name}_{t}/vs_reference_float32", grad[:, :nc], ref, BUDGET)
        assert float(grad[:, nc:].abs().max()) == 0.0 if Zp.shape[1] > nc else True


def test_tsnekhorn_unrolled_matches_the_oracle_at_3000_points():
    """Many tiles / several workgroups: the HIP unrolled gradient against the oracle's dense float64 closed form."""
    import oracle.ref_torch as R
    import torchdr_amd
    from torchdr_amd import _lib
    from torchdr_amd.affinity.entropic import sea_rowstats, sinkhorn_student_adjoint, sinkhorn_student_dual

    n = 3000
    X = gmm(n, 32, 3.0, seed=12)
    sea = torchdr_amd.SymmetricEntropicAffinity(perplexity=20, lr=1e-1, max_iter=15, tol=1e-3, zero_diag=False)
    packed = sea.fit_duals(X.cuda())
    mu, e = sea.dual_side()
    gen = torch.Generator().manual_seed(5)
    Z = torch.randn(n, 2, generator=gen) * 4
    f0 = torch.randn(n, generator=gen) * 0.2
    import oracle
    _, _, C = oracle.knn(X, 0, "sqeuclidean", False, want_full=True)
    mu64, e64 = mu.cpu().double(), e.cpu().double()
    log_P = (mu64[:, None] + mu64[None, :] - 2 * C.double()) / (e64[:, None] + e64[None, :]) - np.log(n)
    ref, f_ref = R.tsnekhorn_unrolled_grad_closed(Z.double(), log_P, f0.double(), 5, 1e-5)
    rec = []
    dual, _ = sinkhorn_student_dual(Z.cuda(), f0.cuda(), 5, 1e-5, True, record=rec)
    assert torch.allclose(dual.cpu().double(), f_ref, rtol=1e-5, atol=1e-5)
    S, _ = sea_rowstats(packed, mu, e, False)
    A, B = sinkhorn_student_adjoint(Z.cuda(), rec, -(2.0 / n) * S, True)
    side = torch.cat([mu[:, None], e[:, None], Z.cuda(), 0.25 * A, B], dim=1).contiguous()
    grad = torch.empty((n, 2), device="cuda")
    _lib.check(_lib.lib().tdr_khorn_grad_unrolled_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(side), 2, float(np.log(n)),
                                                      _lib.ptr(grad), None, 0, _lib.stream_ptr()), "khorn unrolled")
    grade64("tsnekhorn_unrolled/oracle_3000_points_same_duals", grad, ref, BUDGET)


def test_tsnekhorn_four_components_vs_reference():
    """n_components = 4 (no exact instance below 32 other than 2 / 3: the register instance of width 4), duals detached."""
    from torchdr_amd import _lib
    from torchdr_amd.affinity.entropic import sinkhorn_student_dual

    g = load("tsnekhorn_unrolled")
    X = g["X"].cuda()
    n = X.shape[0]
    packed, mu, e = _khorn_side(X)
    init = None
    for t in range(2):
        Z = g[f"n4_Z_{t}"].cuda().contiguous()
        dual, _ = sinkhorn_student_dual(Z, init, 5, 1e-5, True)
        init = dual
        assert torch.allclose(dual.cpu(), g[f"n4_dual_{t}"], rtol=1e-5, atol=2e-6)
        side = torch.cat([mu[:, None], e[:, None], Z, dual.exp()[:, None]], dim=1).contiguous()
        grad = torch.empty((n, 4), device="cuda")
        _lib.check(_lib.lib().tdr_khorn_grad_nc_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(side), 4, float(np.log(n)),
                                                    _lib.ptr(grad), None, 0, _lib.stream_ptr()), "khorn")
        ref = g[f"n4_grad_{t}"]
        grade64(f"tsnekhorn_unrolled/n4_{t}", grad, load("grad64")[f"tsnekhorn_unrolled/n4_grad64_{t}"], BUDGET)
        grade32(f"tsnekhorn_unrolled/n4_{t}/vs_reference_float32", grad, ref, BUDGET)


@pytest.mark.parametrize("kw,name", [(dict(unrolling=True), "u2"), (dict(unrolling=True, n_components=5), "u5"), (dict(n_components=4), "n4")])
def test_tsnekhorn_estimator_steps_vs_reference(kw, name):
    """The estimator itself, two SGD steps from the reference's starting embedding (unrolled / padded widths)."""
    import torchdr_amd

    g = load("tsnekhorn_unrolled")
    rec = {}

    class Probe(torchdr_amd.TSNEkhorn):
        def _init_embedding(self, X):
            super()._init_embedding(X)
            self.embedding_.data.copy_(g[f"{name}_Z_0"].to(self.embedding_.device))

        def _training_step(self):
            t = int(self.n_iter_)
            out = super()._training_step()
            if t < 2:
                rec[t] = self.embedding_.detach().clone()
            return out

    Probe(perplexity=10, max_iter=3, max_iter_affinity_in=30, init="normal", init_scaling=1.0, min_grad_norm=1e-12, lr=1.0,
          optimizer="SGD", optimizer_kwargs=None, random_state=3, **kw).fit_transform(g["X"].cuda())
    for t in range(2):
        ref = g[f"{name}_Zafter_{t}"]
        grade32(f"tsnekhorn_unrolled/{name}_estimator_Zafter_{t}", rec[t], ref, BUDGET)


def test_tsnekhorn_estimator():
    import torchdr_amd

    X = gmm(1500, 32, 4.0, seed=5)
    m = torchdr_amd.TSNEkhorn(perplexity=10, max_iter=60, max_iter_affinity_in=40, init="normal", init_scaling=1.0,
                              lr=1.0, optimizer="SGD", optimizer_kwargs=None, min_grad_norm=1e-12, random_state=0)
    Z = m.fit_transform(X.cuda())
    assert Z.shape == (1500, 2) and torch.isfinite(Z).all()
    # defaults reproduce the reference's behaviour on small inits: the loop stops at iteration 0 because the
    # gradient norm is below min_grad_norm = 1e-4
    m2 = torchdr_amd.TSNEkhorn(perplexity=10, max_iter=50, max_iter_affinity_in=20, random_state=0)
    m2.fit_transform(X.cuda())
    assert int(m2.n_iter_) == 0
    with pytest.raises(ValueError, match="does not support distributed"):
        torchdr_amd.TSNEkhorn(distributed=True)
    with pytest.raises(NotImplementedError, match="fails at its first loss evaluation"):   # so does the reference (RuntimeError)
        torchdr_amd.TSNEkhorn(symmetric_affinity=False)
    Zu = torchdr_amd.TSNEkhorn(perplexity=10, max_iter=20, max_iter_affinity_in=20, init="normal", init_scaling=1.0, lr=1.0,
                               optimizer="SGD", optimizer_kwargs=None, min_grad_norm=1e-12, random_state=0, unrolling=True,
                               n_components=1).fit_transform(X.cuda())
    assert Zu.shape == (1500, 1) and torch.isfinite(Zu).all()


def test_sea_lbfgs_objective_and_run():
    """optimizer="LBFGS" (entropic.py:473-508).  (1) The matrix-free closure -- three row statistics of
    tdr_sea_rowstats3_f32 -- gives the reference's loss and autograd gradients at the fixture's duals (1e-4).  (2) A run
    from the reference's starting point lowers the dual residuals and returns finite duals; the dense log-affinity is
    built from the FINAL duals, as the reference does on this path."""
    import math

    from torchdr_amd.affinity import SymmetricEntropicAffinity
    from torchdr_amd.affinity.entropic import sea_rowstats
    from torchdr_amd.distance import PackedPoints

    g = load("sea_lbfgs")
    X = g["X"].cuda()
    packed = PackedPoints(X)
    target = math.log(10.0) + 1
    for name, sq, zd in (("sq", True, True), ("lin", False, False)):
        eps, mu = g[f"{name}_eps"].cuda(), g[f"{name}_mu"].cuda()
        e = eps ** 2 if sq else eps
        P_sum, H, energy = sea_rowstats(packed, mu, e, zd, energy=True)
        loss = -energy.sum() - torch.inner(e, target - H) + torch.inner(mu, P_sum - 1)
        assert abs(float(loss) - float(g[f"{name}_loss"])) < 3e-4 * abs(float(g[f"{name}_loss"]))
        ge = (2 * eps * (H - target)) if sq else (H - target)
        assert torch.allclose(ge.cpu(), g[f"{name}_grad_eps"], rtol=2e-4, atol=2e-4)
        assert torch.allclose((P_sum - 1).cpu(), g[f"{name}_grad_mu"], rtol=2e-4, atol=1e-4)
    # a run: residuals at the start (eps = mu = 1) against the residuals of the returned duals
    sea = SymmetricEntropicAffinity(perplexity=10, optimizer="LBFGS", lr=1e-3, max_iter=40, tol=1e-4)
    one = torch.ones(256, device="cuda")
    P0, H0 = sea_rowstats(packed, one, one, True)
    r0 = float((P0 - 1).norm() + (H0 - target).norm())
    sea.fit_duals(X)
    mu, e = sea.dual_side()
    assert bool(torch.isfinite(mu).all()) and bool(torch.isfinite(e).all()) and int(sea.n_iter_) > 1
    P1, H1 = sea_rowstats(packed, mu, e, True)
    r1 = float((P1 - 1).norm() + (H1 - target).norm())
    assert r1 < r0, (r0, r1)
    logP = sea(X, log=True)
    assert logP.shape == (256, 256) and bool(torch.isfinite(logP[~torch.eye(256, dtype=torch.bool, device="cuda")]).all())


def test_tsnekhorn_three_components_vs_reference():
    """n_components = 3 (tests/golden/tsnekhorn3.npz): the estimator replayed from the reference's starting embedding --
    Sinkhorn duals, gradients and embeddings of the first two steps."""
    import torchdr_amd

    g = load("tsnekhorn3")
    X = gmm(256, 16, 2.0, seed=61).cuda()
    seen = {}

    class Replay(torchdr_amd.TSNEkhorn):
        def _init_embedding(self, X_):
            self.embedding_ = g["Z_0"].to(self.device_).contiguous()
            return self.embedding_

        def _optimizer_step(self, grad):
            t = int(self.n_iter_)
            if t < 2:
                seen[f"grad_{t}"] = grad.detach().cpu().clone()
                seen[f"dual_{t}"] = self.dual_sinkhorn_.detach().cpu().clone()
            super()._optimizer_step(grad)

        def on_training_step_end(self):
            super().on_training_step_end()
            t = int(self.n_iter_)
            if t < 2:
                seen[t] = self.embedding_.detach().cpu().clone()

    Z = Replay(perplexity=10, n_components=3, max_iter=3, max_iter_affinity_in=30, init="normal", init_scaling=1.0,
               min_grad_norm=1e-12, lr=1.0, optimizer="SGD", optimizer_kwargs=None, random_state=3).fit_transform(X)
    assert Z.shape == (256, 3)
    for t in range(2):
        assert torch.allclose(seen[f"dual_{t}"], g[f"dual_{t}"], rtol=1e-4, atol=1e-5), t
        ref = g[f"grad_{t}"]
        grade64(f"tsnekhorn3/estimator_{t}", seen[f"grad_{t}"], load("grad64")[f"tsnekhorn3/grad64_{t}"], BUDGET)
        grade32(f"tsnekhorn3/estimator_{t}/vs_reference_float32", seen[f"grad_{t}"], ref, BUDGET)
        ref = g[f"Zafter_{t}"]
        grade32(f"tsnekhorn3/estimator_Zafter_{t}", seen[t], ref, BUDGET)
    with pytest.raises(NotImplementedError):
        torchdr_amd.TSNEkhorn(n_components=33)
