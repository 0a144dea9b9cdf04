"""PaCMAP: affinity, closed-form gradient kernel and whole-estimator trajectory against
golden vectors of the real reference (tests/golden/pacmap.npz)."""

import pytest
import torch

from tests.conftest import gmm, grade32, grade64
from tests.test_oracle_golden import load

pytestmark = pytest.mark.gpu


def test_pacmap_affinity_vs_reference():
    from torchdr_amd.affinity import PACMAPAffinity

    g = load("pacmap")
    aff = PACMAPAffinity(n_neighbors=10)
    vals, idx = aff(g["X"].cuda())
    assert vals is None and idx.dtype == torch.int64 and idx.shape == g["aff_idx"].shape
    assert torch.allclose(aff.rho_.cpu(), g["aff_rho"], rtol=1e-6)
    ours, ref = idx.cpu().sort(1).values, g["aff_idx"].sort(1).values
    assert float((ours != ref).any(1).float().mean()) < 0.01  # same neighbour sets (scaled-distance near-ties aside)


def test_pacmap_gradient_vs_reference_autograd():
    from torchdr_amd import _lib

    L = _lib.lib()
    g = load("pacmap")
    n = g["X"].shape[0]
    near = g["pm_NN"].to(torch.int64).cuda().contiguous()
    for t in range(4):
        Z = g[f"pm_Z_{t}"].cuda().contiguous()
        mid = g[f"pm_mid_{t}"].to(torch.int64).cuda().contiguous()
        far = g[f"pm_neg_{t}"].to(torch.int64).cuda().contiguous()
        w = g[f"pm_w_{t}"]
        grad = torch.zeros((n, 2), device="cuda")
        _lib.check(L.tdr_pacmap_grad_f32(_lib.ptr(Z), 2, n, _lib.ptr(near), near.shape[1], float(w[0]), _lib.ptr(mid),
                                         mid.shape[1], float(w[1]), _lib.ptr(far), far.shape[1], float(w[2]),
                                         _lib.ptr(grad), _lib.stream_ptr()), "pacmap_grad")
        ref = g[f"pm_grad_{t}"]
        grade64(f"pacmap/pm_{t}", grad, load("grad64")[f"pacmap/pm_grad64_{t}"], 1e-5)
        grade32(f"pacmap/pm_{t}/vs_reference_float32", grad, ref, 1e-5)


def test_pacmap_estimator_trajectory_vs_reference():
    """Our PACMAP (HIP kNN -> rho rescale -> pair tables -> closed-form gradient -> torch Adam on the device) replaying
    the reference's sampled tables reproduces its embedding through all three weight phases."""
    import torchdr_amd

    g = load("pacmap")
    X = g["X"].cuda()
    seen = {}

    class Replay(torchdr_amd.PACMAP):
        def _init_embedding(self, X_):
            self.embedding_ = g["pm_Z_0"].to(self.device_).contiguous()
            return self.embedding_

        def on_training_step_start(self):
            super().on_training_step_start()
            t = int(self.n_iter_)
            if t < 4:
                self.neg_indices_ = g[f"pm_neg_{t}"]
                self._inject_mid_near = g[f"pm_mid_{t}"]
                w = g[f"pm_w_{t}"]
                assert (float(self.w_NB), float(self.w_MN), float(self.w_FP)) == (float(w[0]), float(w[1]), float(w[2]))
            else:
                self._inject_mid_near = None

        def on_training_step_end(self):
            super().on_training_step_end()
            t = int(self.n_iter_)
            if t < 4:
                seen[t] = self.embedding_.detach().cpu().clone()

    m = Replay(n_neighbors=10, max_iter=5, iter_per_phase=1, random_state=4)
    m.fit_transform(X)
    # the near-pair table is the affinity's output: identical sets as the reference's
    assert torch.equal(m_sorted(g["pm_NN"]), m_sorted(g["aff_idx"]))
    for t in range(4):
        ref = g[f"pm_Zafter_{t}"]
        grade32(f"pacmap/estimator_Zafter_{t}", seen[t], ref, 1e-5)


def m_sorted(t):
    return t.sort(1).values


def test_pacmap_end_to_end():
    import torchdr_amd
    from torchdr_amd.eval import knn_label_accuracy

    n, nc = 3000, 30
    X = gmm(n, 32, 3.0, seed=13).cuda()
    labels = (torch.arange(n) % nc).cuda()
    Z = torchdr_amd.PACMAP(n_neighbors=10, random_state=0).fit_transform(X)
    assert Z.shape == (n, 2) and bool(torch.isfinite(Z).all())
    assert float(knn_label_accuracy(Z, labels, k=10)) > 0.9


@pytest.mark.parametrize("metric,mode", [("sqeuclidean", 0), ("manhattan", 2), ("angular", 3)])
def test_mid_near_kernel_samples_what_the_reference_samples(metric, mode):
    """tdr_pacmap_mid_near_f32 (pacmap.py:213-239 in one launch), through its test hook that emits the picked ROW: every pick is one of six candidates that are uniform over
    the rows the reference can draw (1 .. n - 1, the row itself excluded), and it is the SECOND nearest of them in the input
    space -- checked by recomputing the six candidates' distances from the same hash on the host side of the test: the
    distribution of the pick's rank among fresh uniform rows matches the second order statistic of six."""
    from torchdr_amd import _lib

    n, d, n_mid = 20_000, 24, 5
    X = gmm(n, d, 2.0, seed=3).cuda().contiguous()
    L = _lib.lib()
    out = torch.empty((n, n_mid), dtype=torch.int64, device="cuda")
    _lib.check(L.tdr_pacmap_mid_near_f32(_lib.ptr(X), X.stride(0), d, n, n_mid, mode, 1234567, 7, 1, _lib.ptr(out), _lib.stream_ptr()), "mid")
    out2 = torch.empty_like(out)
    _lib.check(L.tdr_pacmap_mid_near_f32(_lib.ptr(X), X.stride(0), d, n, n_mid, mode, 1234567, 7, 1, _lib.ptr(out2), _lib.stream_ptr()), "mid")
    assert torch.equal(out, out2)                                              # a function of (seed, iteration, row, slot)
    rows = torch.arange(n, device="cuda")[:, None]
    assert int(out.min()) >= 1 and int(out.max()) <= n - 1 and not bool((out == rows).any())
    # uniform over the admissible rows: chi-square of 20 equal bins
    hist = torch.histc(out.float(), bins=20, min=1, max=n - 1)
    exp = out.numel() / 20
    assert float(((hist - exp) ** 2 / exp).sum()) < 60
    # second order statistic of six: the share of uniformly drawn rows that are nearer than the pick has mean 2/7
    def dist(a, b):
        if mode == 0:
            return ((a - b) ** 2).sum(-1)
        if mode == 2:
            return (a - b).abs().sum(-1)
        return -(a * b).sum(-1)
    g = torch.Generator(device="cuda").manual_seed(0)
    probe = torch.randint(1, n, (n, 64), device="cuda", generator=g)
    dp = dist(X[:, None, :], X[probe])                                         # (n, 64)
    dpick = dist(X, X[out[:, 0]])
    share = (dp < dpick[:, None]).float().mean()
    assert abs(float(share) - 2.0 / 7.0) < 0.01, float(share)
    # another iteration / seed draws other pairs
    _lib.check(L.tdr_pacmap_mid_near_f32(_lib.ptr(X), X.stride(0), d, n, n_mid, mode, 1234567, 8, 1, _lib.ptr(out2), _lib.stream_ptr()), "mid")
    assert float((out2 == out).float().mean()) < 0.01
    # production form: the POSITION of that candidate among the six (the reference's `topk(...).indices[:, 1]`): 0..5, uniform
    pos = torch.empty_like(out)
    _lib.check(L.tdr_pacmap_mid_near_f32(_lib.ptr(X), X.stride(0), d, n, n_mid, mode, 1234567, 7, 0, _lib.ptr(pos), _lib.stream_ptr()), "mid")
    assert int(pos.min()) == 0 and int(pos.max()) == 5
    h6 = torch.bincount(pos.reshape(-1), minlength=6).float()
    assert float(((h6 - pos.numel() / 6) ** 2 / (pos.numel() / 6)).sum()) < 30
