"""TSNEkhorn on MI355X -- mirror of ``torchdr/neighbor_embedding/tsnekhorn.py`` (reference :110-230)."""

import math
from typing import Dict, Optional, Type, Union

import torch

from torchdr_amd import _lib
from torchdr_amd.affinity.entropic import (DensePoints64, SinkhornAffinity, SymmetricEntropicAffinity, pad_embedding, pair_scan_workspace,
                                           pairs64_workspace, sea_rowstats, sinkhorn_student_adjoint, sinkhorn_student_dual)
from torchdr_amd.neighbor_embedding.base import NeighborEmbedding
from torchdr_amd.utils import bool_arg


class TSNEkhorn(NeighborEmbedding):
    r"""SNEkhorn with a Student-t output kernel: symmetric entropic affinity in, symmetric Sinkhorn
    affinity out, loss :math:`\mathrm{CE}(P, \log Q) + \sum Q` (reference ``tsnekhorn.py:210-230``).

    Fully matrix-free: the input affinity lives as its duals :math:`(\varepsilon, \mu)` and the packed
    point images; every training step recomputes :math:`P_{ij}` tile by tile on the MFMA pipe inside the
    fused force kernel (``tdr_khorn_grad_nc_f32``), after 5 warm-started Sinkhorn passes on the embedding
    (``tdr_sinkhorn_pass_f32``).  Any ``n_components`` up to 32 (zero-padded register instances).

    ``unrolling=True`` (reference ``tsnekhorn.py:134, 183, 224-227``: the loss is :math:`\mathrm{CE}(P, \log Q)` alone and
    autograd runs THROUGH the 5 Sinkhorn updates) is differentiated in closed form: the forward passes record
    :math:`(e^{f^{k-1}}, 1/s^k)`, a reverse sweep of 5 Student-kernel mat-vecs gives the adjoints
    (``affinity.entropic.sinkhorn_student_adjoint``), and the force kernel ``tdr_khorn_grad_unrolled_f32`` carries them as a
    rank-10 bilinear form in place of the :math:`Q` term -- still nothing of size :math:`N^2`.

    ``symmetric_affinity=False`` is refused: the reference's own path fails at its first loss evaluation (the sparse
    ``(n, k)`` entropic affinity is multiplied with the dense ``(n, n)`` log Q: "The size of tensor a (k) must match the size of
    tensor b (n)"), so there is no behaviour to reproduce."""

    # float64 inputs are embedded in float64 (csrc/tdr_khorn_f64.hip: dense float64 distance matrix, n <= 16384, 2-4 components)
    _float64_loop = True

    def _float64_ok(self, X) -> bool:
        return 2 <= int(self.n_components) <= 4 and DensePoints64.eligible(X) and self.metric == "sqeuclidean"

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
        if not self.symmetric_affinity:
            raise NotImplementedError(
                "[torchdr_amd] TSNEkhorn(symmetric_affinity=False): the reference's own path fails at its first loss evaluation "
                "(sparse (n, k) affinity against the dense (n, n) log Q); there is no behaviour to reproduce."
            )
        if n_components > 32:
            raise NotImplementedError("[torchdr_amd] TSNEkhorn: n_components above 32 is not part of the accelerated path.")
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
        self._p_marginals = None

    def _compute_gradients64(self):
        """float64 form of the step below: the same five warm-started Sinkhorn updates and the same force, reduced by the float64
        kernels over the dense float64 distance matrix of the input (``DensePoints64``)."""
        n, nc = self.n_samples_in_, self.n_components
        Z = self.embedding_.detach()
        out = self.affinity_out
        rec = [] if self.unrolling else None
        dual, k = sinkhorn_student_dual(Z, self.dual_sinkhorn_, out.max_iter, out.tol, out.zero_diag, record=rec)
        out.register_buffer("dual_", dual, persistent=False)
        out.n_iter_ = k
        self.dual_sinkhorn_ = dual.detach()
        grad = torch.empty((n, nc), dtype=torch.float64, device=self.device_)
        ws, ws_bytes = pairs64_workspace(n, nc, self.device_)
        L, C = _lib.lib(), self._packed.C
        if not self.unrolling:
            side = torch.cat([self._mu[:, None], self._e[:, None], Z, dual.exp()[:, None]], dim=1).contiguous()
            _lib.check(L.tdr_khorn_grad_dense_f64(_lib.ptr(C), n, C.stride(0), _lib.ptr(side), nc, math.log(n), _lib.ptr(grad), _lib.ptr(ws),
                                                  ws_bytes, _lib.stream_ptr()), "tdr_khorn_grad_dense_f64")
        else:
            if self._p_marginals is None:
                S, _ = sea_rowstats(self._packed, self._mu, self._e, False)
                self._p_marginals = (2.0 / n) * S
            A, B = sinkhorn_student_adjoint(Z, rec, -self._p_marginals, out.zero_diag)
            side = torch.cat([self._mu[:, None], self._e[:, None], Z, 0.25 * A, B], dim=1).contiguous()
            _lib.check(L.tdr_khorn_grad_unrolled_dense_f64(_lib.ptr(C), n, C.stride(0), _lib.ptr(side), nc, math.log(n), _lib.ptr(grad),
                                                           _lib.ptr(ws), ws_bytes, _lib.stream_ptr()), "tdr_khorn_grad_unrolled_dense_f64")
        return grad, False

    def _compute_gradients(self):
        if isinstance(self._packed, DensePoints64):
            return self._compute_gradients64()
        n = self.n_samples_in_
        nc = self.n_components
        Zp = pad_embedding(self.embedding_)
        w = Zp.shape[1]
        out = self.affinity_out
        rec = [] if self.unrolling else None
        # 5 warm-started passes from the detached dual of the previous step (:214-216)
        dual, k = sinkhorn_student_dual(Zp, self.dual_sinkhorn_, out.max_iter, out.tol, out.zero_diag, record=rec)
        out.register_buffer("dual_", dual, persistent=False)
        out.n_iter_ = k
        self.dual_sinkhorn_ = dual.detach()
        grad = torch.empty((n, w), dtype=torch.float32, device=self.device_)
        ws, ws_bytes, _keep = pair_scan_workspace(n, w, self.device_)
        L = _lib.lib()
        if not self.unrolling:
            side = torch.cat([self._mu[:, None], self._e[:, None], Zp, dual.exp()[:, None]], dim=1).contiguous()
            _lib.check(L.tdr_khorn_grad_nc_f32(_lib.ptr(self._packed.data), n, self._packed.d, _lib.ptr(side), w, math.log(n),
                                               _lib.ptr(grad), ws, ws_bytes, _lib.stream_ptr()), "tdr_khorn_grad_nc_f32")
        else:
            if self._p_marginals is None:    # d loss / d f_i = -(sum_j P_ij + sum_j P_ji); P is fixed during the fit
                S, _ = sea_rowstats(self._packed, self._mu, self._e, False)
                self._p_marginals = (2.0 / n) * S
            A, B = sinkhorn_student_adjoint(Zp, rec, -self._p_marginals, out.zero_diag)
            side = torch.cat([self._mu[:, None], self._e[:, None], Zp, 0.25 * A, B], dim=1).contiguous()
            _lib.check(L.tdr_khorn_grad_unrolled_f32(_lib.ptr(self._packed.data), n, self._packed.d, _lib.ptr(side), w,
                                                     math.log(n), _lib.ptr(grad), ws, ws_bytes, _lib.stream_ptr()),
                       "tdr_khorn_grad_unrolled_f32")
        return (grad if w == nc else grad[:, :nc].contiguous()), False

    def clear_memory(self):
        super().clear_memory()
        for attr in ("_packed", "_mu", "_e", "dual_sinkhorn_", "_p_marginals"):
            if hasattr(self, attr):
                delattr(self, attr)
