"""Riemannian Adam (reference ``utils/radam.py:57-187``), twice:

* ``PoincareAdamKernel`` -- what COSNE steps with: the HIP kernel ``tdr_radam_poincare_f64`` (one thread per row: egrad2rgrad,
  the two moments, expmap, projection and the parallel transport of the first moment, ``utils/manifold.py:207-330``) plus
  the hyper-parameters and the state it needs;
* ``RiemannianAdam`` -- the reference's ``torch.optim.Adam`` subclass for arbitrary parameter lists (``ManifoldParameter``s
  on their manifold, everything else Euclidean), written with torch ops on the classes of ``utils/manifold.py``: the name
  user code imports.  Same quirks: the group's step counter advances twice per parameter step (:147, :163) and the
  squared-gradient moment holds the metric inner product of the whole row."""

import torch

from torchdr_amd import _lib

_MAXNORM_F64 = 1.0 - 1e-5      # PoincareBallManifold.eps[float64] (manifold.py:214, 233)


class PoincareAdamKernel:
    """State + hyper-parameters of the reference's RiemannianAdam on the unit Poincare ball (c = 1), stepped by the kernel."""

    def __init__(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, stabilize=None):
        if weight_decay != 0.0 or amsgrad or stabilize is not None:
            raise NotImplementedError("[torchdr_amd] COSNE's fused Riemannian-Adam step: weight_decay / amsgrad / stabilize are not built.")
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


def _keep_strides(dest, source):
    """Write `source` into the user's tensor without changing how it is laid out."""
    return dest.copy_(source) if dest.stride() != source.stride() else dest.set_(source)


class RiemannianAdam(torch.optim.Adam):
    """Adam on manifolds with ``torch.optim.Adam``'s interface; ``stabilize=k`` re-projects the manifold parameters every
    time the group's step counter is a multiple of k."""

    def __init__(self, *args, stabilize=None, **kwargs):
        self._stabilize = stabilize
        super().__init__(*args, **kwargs)

    def stabilize(self):
        for group in self.param_groups:
            self.stabilize_group(group)

    @torch.no_grad()
    def stabilize_group(self, group):
        from torchdr_amd.utils.manifold import ManifoldParameter

        for p in group["params"]:
            state = self.state[p] if isinstance(p, ManifoldParameter) else None
            if not state:       # a plain tensor, or a parameter that has not seen a gradient yet
                continue
            _keep_strides(p, p.manifold.proj(p, p.c))
            state["exp_avg"].set_(p.manifold.proj_tan(state["exp_avg"], p, p.c))

    def step(self, closure=None):
        from torchdr_amd.utils.manifold import EuclideanManifold, ManifoldParameter

        loss = closure() if closure is not None else None
        flat = EuclideanManifold()
        with torch.no_grad():
            for group in self.param_groups:
                group.setdefault("step", 0)
                b1, b2 = group["betas"]
                for point in group["params"]:
                    grad = point.grad
                    if grad is None:
                        continue
                    on_manifold = isinstance(point, ManifoldParameter)
                    manifold, c = (point.manifold, point.c) if on_manifold else (flat, None)
                    if grad.is_sparse:
                        raise RuntimeError("Riemannian Adam does not support sparse gradients yet (PR is welcome)")
                    state = self.state[point]
                    if not state:
                        state["step"] = 0
                        state["exp_avg"] = torch.zeros_like(point)
                        state["exp_avg_sq"] = torch.zeros_like(point)
                        if group["amsgrad"]:
                            state["max_exp_avg_sq"] = torch.zeros_like(point)
                    m1, m2 = state["exp_avg"], state["exp_avg_sq"]
                    grad.add_(point, alpha=group["weight_decay"])
                    grad = manifold.egrad2rgrad(point, grad, c)
                    m1.mul_(b1).add_(grad, alpha=1 - b1)
                    m2.mul_(b2).add_(manifold.inner(point, c, grad, keepdim=True), alpha=1 - b2)
                    if group["amsgrad"]:
                        seen = state["max_exp_avg_sq"]
                        torch.max(seen, m2, out=seen)
                        denom = seen.sqrt().add_(group["eps"])
                    else:
                        denom = m2.sqrt().add_(group["eps"])
                    group["step"] += 1
                    t = group["step"]
                    step_size = group["lr"] * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
                    moved = manifold.proj(manifold.expmap(-step_size * (m1 / denom), point, c), c)
                    carried = manifold.ptransp(point, moved, m1, c)     # the first moment travels with the point
                    _keep_strides(point, moved)
                    m1.set_(carried)
                    group["step"] += 1
                if self._stabilize is not None and group["step"] % self._stabilize == 0:
                    self.stabilize_group(group)
        return loss
