"""PaCMAP on MI355X -- mirror of ``torchdr/neighbor_embedding/pacmap.py`` (reference :93-265)."""

from typing import Any, Dict, Optional, Type, Union

import torch

from torchdr_amd import _lib
from torchdr_amd.affinity import PACMAPAffinity
from torchdr_amd.distance import pairwise_distances_indexed
from torchdr_amd.neighbor_embedding.base import NegativeSamplingNeighborEmbedding
from torchdr_amd.utils import compute_device


class PACMAP(NegativeSamplingNeighborEmbedding):
    r"""PaCMAP: near pairs from :class:`PACMAPAffinity`, mid-near pairs re-sampled every iteration while their
    weight is non-zero, ``n_further`` uniformly sampled further pairs (nearest neighbours excluded,
    ``discard_NNs=True``), and the three-phase weight schedule of the reference (``_set_weights``, :182-198).
    The loss gradient is evaluated in closed form by ``tdr_pacmap_grad_f32``; the optimizer is the reference's
    default Adam (``torch.optim`` on the device embedding).

    Faithful to the reference in one detail worth knowing: ``mid_near_indices[:, i]`` stores ``kmin(D, 2)[1][:, 1]``
    (:238-239), i.e. the POSITION (0..5) of the second-closest of the six sampled candidates, not the candidate's
    index -- so the mid-near partners are the points 0..5.  Reproduced as is for result parity."""

    _float64_loop = True   # float64 inputs are embedded in float64 (tdr_pacmap_grad_f64; the mid-near table through the torch form)

    def __init__(self, n_neighbors: float = 10, n_components: int = 2, lr: Union[float, str] = 1e0,
                 optimizer: Union[str, Type[torch.optim.Optimizer]] = "Adam",
                 optimizer_kwargs: Optional[Union[Dict, str]] = None,
                 scheduler: Optional[Union[str, Type[torch.optim.lr_scheduler.LRScheduler]]] = None,
                 scheduler_kwargs: Optional[Dict] = None, init: str = "pca", init_scaling: float = 1e-4,
                 min_grad_norm: float = 1e-7, max_iter: int = 450, device: str = "auto", backend="faiss",
                 verbose: bool = False, random_state: Optional[float] = None, metric: str = "sqeuclidean",
                 MN_ratio: float = 0.5, FP_ratio: float = 2, check_interval: int = 50, iter_per_phase: int = 100,
                 discard_NNs: bool = True, compile: bool = False, distributed: Union[bool, str] = False, **kwargs):
        if distributed:
            raise ValueError("[TorchDR] ERROR : PACMAP does not support distributed.")
        self.n_neighbors = n_neighbors
        self.metric = metric
        self.MN_ratio = MN_ratio
        self.FP_ratio = FP_ratio
        self.n_mid_near = int(MN_ratio * n_neighbors)
        self.n_further = int(FP_ratio * n_neighbors)
        self.iter_per_phase = iter_per_phase
        affinity_in = PACMAPAffinity(n_neighbors=n_neighbors, metric=metric, device=device, backend=backend,
                                     verbose=verbose)
        super().__init__(affinity_in=affinity_in, n_components=n_components, optimizer=optimizer,
                         optimizer_kwargs=optimizer_kwargs, min_grad_norm=min_grad_norm, max_iter=max_iter, lr=lr,
                         scheduler=scheduler, scheduler_kwargs=scheduler_kwargs, init=init,
                         init_scaling=init_scaling, device=device, backend=backend, verbose=verbose,
                         random_state=random_state, check_interval=check_interval, n_negatives=self.n_further,
                         discard_NNs=discard_NNs, compile=compile, distributed=distributed, **kwargs)

    def _fit_transform(self, X: torch.Tensor, y: Optional[Any] = None):
        self.X_ = X.to(compute_device(X, self.device))  # mid-near candidates are ranked in the input space
        self.mid_near_indices = None
        self._inject_mid_near = None  # tests: a (n, n_mid_near) table to use instead of sampling
        self._set_weights()  # with the pre-loop n_iter_ (-1), as the reference does (:171)
        return super()._fit_transform(X, y)

    def _set_weights(self):
        t, T = int(self.n_iter_), self.iter_per_phase
        if t < T:
            self.w_NB, self.w_MN, self.w_FP = 2, 1000 * (1 - t / T) + 3 * t / T, 1
        elif t < 2 * T:
            self.w_NB, self.w_MN, self.w_FP = 3, 3, 1
        else:
            self.w_NB, self.w_MN, self.w_FP = 1, 0, 1

    def on_training_step_end(self):
        super().on_training_step_end()
        self._set_weights()

    def _sample_mid_near(self):
        """pacmap.py:213-239: for each of the n_mid_near slots draw 6 candidates (self excluded), rank them by their
        INPUT-space distance, keep position [1] of the ascending order (see the class docstring)."""
        n = self.n_samples_in_
        if n - 1 < 6:
            raise ValueError("[TorchDR] ERROR : Not enough points to sample 6 mid-near points.")
        dev = self.device_
        mode = {"sqeuclidean": 0, "euclidean": 0, "manhattan": 2, "angular": 3}.get(self.metric)
        if mode is not None and self.X_.dtype == torch.float32 and self.X_.is_cuda and n >= 8:
            # one launch: candidates from the counter hash, their input-space distances, the POSITION of the second nearest
            # among the six (what `topk(...).indices[:, 1]` below stores -- the reference's quirk, class docstring)
            X = self.X_ if self.X_.stride(1) == 1 else self.X_.contiguous()
            out = torch.empty((n, self.n_mid_near), dtype=torch.int64, device=dev)
            _lib.check(_lib.lib().tdr_pacmap_mid_near_f32(_lib.ptr(X), X.stride(0), X.shape[1], n, self.n_mid_near, mode,
                                                          int(self._neg_seed), int(self.n_iter_), 0, _lib.ptr(out), _lib.stream_ptr()),
                       "tdr_pacmap_mid_near_f32")
            return out
        self_idx = torch.arange(n, device=dev).unsqueeze(1)
        out = torch.empty((n, self.n_mid_near), dtype=torch.int64, device=dev)
        for i in range(self.n_mid_near):
            cand = torch.randint(1, n - 1, (n, 6), device=dev)
            cand = cand + torch.searchsorted(self_idx, cand, right=True)
            D = pairwise_distances_indexed(self.X_, key_indices=cand, metric=self.metric)
            out[:, i] = torch.topk(D, 2, dim=1, largest=False).indices[:, 1]
        return out

    def _compute_gradients(self):
        n, nc = self.n_samples_in_, self.n_components
        dt = self.embedding_.dtype
        grad = torch.zeros((n, nc), dtype=dt, device=self.device_)
        near = self.NN_indices_.to(torch.int64).contiguous()
        mid = None
        if self.w_MN > 0:
            mid = self._inject_mid_near if self._inject_mid_near is not None else self._sample_mid_near()
            mid = mid.to(device=self.device_, dtype=torch.int64).contiguous()
            self.mid_near_indices = mid
        far = self._neg_ptr_tensor()
        if far is None:
            raise RuntimeError("[torchdr_amd] PACMAP needs the further-pair table (discard_NNs=True path).")
        lam, rho = float(self.early_exaggeration_coeff_), float(self.repulsion_strength)
        _lib.check(
            _lib.fn("tdr_pacmap_grad", dt)(
                _lib.ptr(self.embedding_), nc, n, _lib.ptr(near), near.shape[1], lam * float(self.w_NB),
                _lib.ptr(mid), 0 if mid is None else mid.shape[1], lam * float(self.w_MN),
                _lib.ptr(far), far.shape[1], rho * float(self.w_FP), _lib.ptr(grad), _lib.stream_ptr(),
            ),
            "tdr_pacmap_grad",
        )
        return grad, False

    def on_training_step_start(self):
        super().on_training_step_start()
        if self._exclusion is None:  # discard_NNs=False: plain uniform negatives, self excluded (base.py:628-636)
            r = torch.randint(0, self.n_samples_in_ - 1, (self.chunk_size_, self.n_negatives), device=self.device_)
            self.neg_indices_ = r + (r >= self.chunk_indices_.unsqueeze(1)).long()

    def clear_memory(self):
        super().clear_memory()
        for attr in ("X_", "mid_near_indices", "_inject_mid_near"):
            if hasattr(self, attr):
                delattr(self, attr)
