"""UMAP on MI355X -- mirror of ``torchdr/neighbor_embedding/umap.py`` (reference :19-292)."""

from typing import Dict, Optional, Type, Union

import numpy as np
import torch

from torchdr_amd import _lib
from torchdr_amd.affinity import UMAPAffinity
from torchdr_amd.neighbor_embedding.base import NegativeSamplingNeighborEmbedding
from torchdr_amd.utils.sparse import CSRAffinity


# bench.py sets this to a list to collect (start_event, end_event, nnz) around every PROFILE_EVERY-th gradient
# evaluation (HIP events on the launch stream); None = no instrumentation.
PROFILE = None
PROFILE_EVERY = 25


def find_ab_params(spread, min_dist):
    """Fit a, b of 1/(1 + a x^(2b)) to the smooth-step target curve (reference :19-36: same grid,
    scipy ``curve_fit``; defaults give a = 1.5769..., b = 0.8950...)."""
    from scipy.optimize import curve_fit

    def curve(x, a, b):
        return 1.0 / (1.0 + a * x ** (2 * b))

    xv = np.linspace(0, spread * 3, 300)
    yv = np.zeros(xv.shape)
    yv[xv < min_dist] = 1.0
    yv[xv >= min_dist] = np.exp(-(xv[xv >= min_dist] - min_dist) / spread)
    params, _ = curve_fit(curve, xv, yv)
    return params[0].item(), params[1].item()


class UMAP(NegativeSamplingNeighborEmbedding):
    """UMAP with the reference's constructor (``umap.py:129-160``) and semantics (appendix A.3 of
    SURVEY.md): squared-Euclidean kNN, sigma search, fuzzy-union symmetrisation, per-edge epoch
    counters, 5 negatives per active edge, forces clamped to [-4, 4], plain SGD with a linear
    1 -> 0 learning-rate ramp."""

    def __init__(self, n_neighbors: float = 30, n_components: int = 2, min_dist: float = 0.1, spread: float = 1.0,
                 a: Optional[float] = None, b: Optional[float] = None, lr: float = 1e0,
                 optimizer: Union[str, Type[torch.optim.Optimizer]] = "SGD",
                 optimizer_kwargs: Union[Dict, str] = None,
                 scheduler: Optional[Union[str, Type[torch.optim.lr_scheduler.LRScheduler]]] = "LinearLR",
                 scheduler_kwargs: Union[Dict, str, None] = "auto", init: str = "pca", init_scaling: float = 1e-4,
                 min_grad_norm: float = 1e-7, max_iter: int = 1000, device: str = "auto", backend="faiss",
                 verbose: bool = False, random_state: Optional[float] = None, max_iter_affinity: int = 100,
                 metric: str = "sqeuclidean", negative_sample_rate: int = 5, check_interval: int = 50,
                 discard_NNs: bool = False, compile: bool = False, distributed: Union[bool, str] = "auto",
                 **kwargs):
        self.n_neighbors = n_neighbors
        self.min_dist = min_dist
        self.spread = spread
        self.metric = metric
        self.max_iter_affinity = max_iter_affinity
        self.negative_sample_rate = negative_sample_rate
        self.sparsity = True
        self._use_closed_form_gradients = True
        self._eps = 1e-3
        self.a, self.b = a, b  # as given (sklearn get_params); the fitted curve parameters live in _a / _b
        if a is None or b is None:
            a, b = find_ab_params(self.spread, self.min_dist)
        self._a = a
        self._b = b
        self.n_negatives = int(self.negative_sample_rate * self.n_neighbors)
        affinity_in = UMAPAffinity(n_neighbors=n_neighbors, metric=metric, max_iter=max_iter_affinity,
                                   device=device, backend=backend, verbose=verbose, sparsity=self.sparsity,
                                   compile=compile, distributed=distributed)
        super().__init__(affinity_in=affinity_in, n_components=n_components, optimizer=optimizer,
                         optimizer_kwargs=optimizer_kwargs, min_grad_norm=min_grad_norm, max_iter=max_iter, lr=lr,
                         scheduler=scheduler, scheduler_kwargs=scheduler_kwargs, init=init,
                         init_scaling=init_scaling, device=device, backend=backend, verbose=verbose,
                         random_state=random_state, check_interval=check_interval, discard_NNs=discard_NNs,
                         compile=compile, n_negatives=self.n_negatives, distributed=distributed, **kwargs)

    # the affinity stays in CSR on the device (the reference's padded (N, max_deg) layout is 5-8x larger)
    def _compute_affinity_in(self, X):
        self._csr = self.affinity_in(X, return_indices=True, return_csr=True)

    def _nn_for_exclusion(self):
        _, idx = self._csr.to_padded()
        return idx

    def on_affinity_computation_end(self):
        super().on_affinity_computation_end()
        csr: CSRAffinity = self._csr
        self.epochs_per_sample = torch.empty_like(csr.vals)
        self.epoch_of_next_sample = torch.empty_like(csr.vals)
        scratch = torch.zeros(2, dtype=torch.int32, device=csr.vals.device)
        _lib.check(
            _lib.lib().tdr_umap_prepare_f32(_lib.ptr(csr.vals), csr.nnz, int(self.max_iter),
                                            _lib.ptr(self.epochs_per_sample), _lib.ptr(self.epoch_of_next_sample),
                                            _lib.ptr(scratch), _lib.stream_ptr()),
            "tdr_umap_prepare_f32",
        )

    def _compute_gradients(self):
        csr: CSRAffinity = self._csr
        if self.world_size > 1 or getattr(self, "_grad_buf", None) is None:
            self._grad_buf = torch.empty((self.chunk_size_, self.n_components), dtype=torch.float32,
                                         device=self.device_)
        grad = self._grad_buf
        neg = self._neg_ptr_tensor()
        if getattr(self, "_grad_ws", None) is None:  # scratch of the L2-sliced negative phase (large N only)
            nbytes = _lib.lib().tdr_umap_grad_workspace_bytes(self.n_samples_in_, self.chunk_size_, self.n_components)
            self._grad_ws = torch.empty(max(nbytes, 8) // 4 + 1, dtype=torch.int32, device=self.device_)
            self._grad_ws_bytes = nbytes
        prof = PROFILE is not None and int(self.n_iter_) % PROFILE_EVERY == 0
        if prof:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        _lib.check(
            _lib.lib().tdr_umap_grad_f32(
                _lib.ptr(self.embedding_), self.n_components, self.n_samples_in_, self.chunk_start_,
                self.chunk_size_, _lib.ptr(csr.rowptr), _lib.ptr(csr.cols), _lib.ptr(self.epochs_per_sample),
                _lib.ptr(self.epoch_of_next_sample), float(self._a), float(self._b), int(self.n_iter_),
                int(self.negative_sample_rate), int(self.n_negatives), _lib.ptr(neg), self._neg_seed,
                float(self.early_exaggeration_coeff_), float(self.repulsion_strength), float(self._eps),
                _lib.ptr(grad), 0, _lib.ptr(self._grad_ws), self._grad_ws_bytes, _lib.stream_ptr(),
            ),
            "tdr_umap_grad_f32",
        )
        if prof:
            ev1.record()
            PROFILE.append((ev0, ev1, csr.nnz))
        return grad, True

    def clear_memory(self):
        super().clear_memory()
        for attr in ("_csr", "epochs_per_sample", "epoch_of_next_sample", "_exclusion", "_grad_buf", "_grad_ws"):
            if hasattr(self, attr):
                delattr(self, attr)
