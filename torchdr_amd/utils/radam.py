"""``RiemannianAdam`` -- the optimizer name COSNE is configured with (reference ``utils/radam.py:57-167``).

On this path the step itself is the HIP kernel ``tdr_radam_poincare_f64`` (one thread per row: egrad2rgrad, the two
moments, expmap, projection and the parallel transport of the first moment, ``utils/manifold.py:207-330``); this class
carries the hyper-parameters and the state."""

import torch

from torchdr_amd import _lib

_MAXNORM_F64 = 1.0 - 1e-5      # PoincareBallManifold.eps[float64] (manifold.py:214, 233)


class RiemannianAdam:
    """State + hyper-parameters of the reference's RiemannianAdam on the unit Poincare ball (c = 1)."""

    def __init__(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, stabilize=None):
        if weight_decay != 0.0 or amsgrad or stabilize is not None:
            raise NotImplementedError("[torchdr_amd] RiemannianAdam: weight_decay / amsgrad / stabilize are not built.")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.step_count = 0            # the reference increments its group counter TWICE per step (radam.py:147, 163)
        self.exp_avg = None
        self.exp_avg_sq = None

    def step(self, rows: torch.Tensor, egrad: torch.Tensor, lr=None, nan_flag=None, n_iter=0, want_rgrad=True):
        """Update ``rows`` (a contiguous float64 view of the embedding) in place; returns the Riemannian gradient."""
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(rows)
            self.exp_avg_sq = torch.zeros_like(rows)
        self.step_count += 1
        t = self.step_count
        b1, b2 = self.betas
        step_size = (self.lr if lr is None else float(lr)) * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
        rgrad = torch.empty_like(rows) if want_rgrad else None
        _lib.check(
            _lib.lib().tdr_radam_poincare_f64(_lib.ptr(rows), _lib.ptr(egrad), _lib.ptr(self.exp_avg),
                                              _lib.ptr(self.exp_avg_sq), _lib.ptr(rgrad), rows.shape[0], rows.shape[1],
                                              b1, b2, self.eps, step_size, _MAXNORM_F64, _lib.ptr(nan_flag), int(n_iter),
                                              _lib.stream_ptr()),
            "tdr_radam_poincare_f64",
        )
        self.step_count += 1
        return rgrad
