"""PaCMAP: affinity, closed-form gradient kernel and whole-estimator trajectory against
golden vectors of the real reference (tests/golden/pacmap.npz)."""

import pytest
import torch

from tests.conftest import gmm
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
        assert torch.allclose(grad.cpu(), ref, rtol=1e-4, atol=2e-6 * float(ref.abs().max())), f"step {t}"


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
        assert torch.allclose(seen[t], ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max())), f"step {t}"


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
