"""TSNEkhorn on MI355X -- mirror of ``torchdr/neighbor_embedding/tsnekhorn.py`` (reference :110-230)."""

import math
from typing import Dict, Optional, Type, Union

import torch

from torchdr_amd import _lib
from torchdr_amd.affinity.entropic import SinkhornAffinity, SymmetricEntropicAffinity
from torchdr_amd.neighbor_embedding.base import NeighborEmbedding
from torchdr_amd.utils import bool_arg


class TSNEkhorn(NeighborEmbedding):
    r"""SNEkhorn with a Student-t output kernel: symmetric entropic affinity in, symmetric Sinkhorn
    affinity out, loss :math:`\mathrm{CE}(P, \log Q) + \sum Q` (reference ``tsnekhorn.py:210-230``).

    Fully matrix-free: the input affinity lives as its duals :math:`(\varepsilon, \mu)` and the packed
    point images; every training step recomputes :math:`P_{ij}` tile by tile on the MFMA pipe inside the
    fused force kernel (``tdr_khorn_grad_f32``), after 5 warm-started Sinkhorn passes on the embedding
    (``tdr_sinkhorn_pass_f32``).  ``unrolling`` (autograd through the Sinkhorn loop) and the non-symmetric
    input affinity are not part of the accelerated path."""

    def __init__(self, perplexity: float = 30, n_components: int = 2, lr: Union[float, str] = "auto",
                 optimizer: Union[str, Type[torch.optim.Optimizer]] = "SGD",
                 optimizer_kwargs: Union[Dict, str] = "auto",
                 scheduler: Optional[Union[str, Type[torch.optim.lr_scheduler.LRScheduler]]] = None,
                 scheduler_kwargs: Optional[Dict] = None, init: str = "pca", init_scaling: float = 1e-4,
                 min_grad_norm: float = 1e-4, max_iter: int = 2000, device: str = "auto", backend=None,
                 verbose: bool = False, random_state: Optional[float] = None, lr_affinity_in: float = 1e-1,
                 eps_square_affinity_in: bool = True, tol_affinity_in: float = 1e-3,
                 max_iter_affinity_in: int = 100, metric: str = "sqeuclidean", unrolling: bool = False,
                 symmetric_affinity: bool = True, check_interval: int = 50, compile: bool = False,
                 distributed: Union[bool, str] = False, **kwargs):
        if distributed:
            raise ValueError("[TorchDR] ERROR : TSNEkhorn does not support distributed.")
        self.metric = metric
        self.perplexity = perplexity
        self.lr_affinity_in = lr_affinity_in
        self.eps_square_affinity_in = bool_arg(eps_square_affinity_in)
        self.max_iter_affinity_in = max_iter_affinity_in
        self.tol_affinity_in = tol_affinity_in
        self.unrolling = bool_arg(unrolling)
        self.symmetric_affinity = bool_arg(symmetric_affinity)
        if self.unrolling or not self.symmetric_affinity or n_components not in (2, 3):
            raise NotImplementedError(
                "[torchdr_amd] TSNEkhorn: unrolling=True, symmetric_affinity=False and n_components outside {2, 3} "
                "are not part of the accelerated path."
            )
        affinity_in = SymmetricEntropicAffinity(perplexity=perplexity, lr=lr_affinity_in,
                                                eps_square=eps_square_affinity_in, metric=metric,
                                                tol=tol_affinity_in, max_iter=max_iter_affinity_in, device=device,
                                                backend=backend, verbose=verbose, zero_diag=False)
        affinity_out = SinkhornAffinity(metric="sqeuclidean", device=device, backend=backend, verbose=False,
                                        base_kernel="student", with_grad=unrolling, max_iter=5)
        super().__init__(affinity_in=affinity_in, affinity_out=affinity_out, n_components=n_components,
                         optimizer=optimizer, optimizer_kwargs=optimizer_kwargs, min_grad_norm=min_grad_norm,
                         max_iter=max_iter, lr=lr, scheduler=scheduler, scheduler_kwargs=scheduler_kwargs,
                         init=init, init_scaling=init_scaling, device=device, backend=backend, verbose=verbose,
                         random_state=random_state, check_interval=check_interval, compile=compile,
                         distributed=distributed, **kwargs)

    def _compute_affinity_in(self, X):
        self._packed = self.affinity_in.fit_duals(X)
        self._mu, self._e = self.affinity_in.dual_side()
        self.dual_sinkhorn_ = None

    def _compute_gradients(self):
        n = self.n_samples_in_
        Z = self.embedding_.detach()
        dual = self.affinity_out.fit_dual(Z, init_dual=self.dual_sinkhorn_)  # 5 warm-started passes (:214-216)
        self.dual_sinkhorn_ = dual.detach()
        nc = self.n_components
        side = torch.cat([self._mu[:, None], self._e[:, None], Z, dual.exp()[:, None]], dim=1).contiguous()
        grad = torch.empty((n, nc), dtype=torch.float32, device=self.device_)
        _lib.check(
            _lib.lib().tdr_khorn_grad_nc_f32(_lib.ptr(self._packed.data), n, self._packed.d, _lib.ptr(side), nc,
                                             math.log(n), _lib.ptr(grad), _lib.stream_ptr()),
            "tdr_khorn_grad_nc_f32",
        )
        return grad, False

    def clear_memory(self):
        super().clear_memory()
        for attr in ("_packed", "_mu", "_e", "dual_sinkhorn_"):
            if hasattr(self, attr):
                delattr(self, attr)
