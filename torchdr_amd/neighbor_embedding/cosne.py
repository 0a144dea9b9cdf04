"""CO-SNE on MI355X -- mirror of ``torchdr/neighbor_embedding/cosne.py`` (reference :94-193)."""

from typing import Any, Dict, Optional, Type, Union

import torch

from torchdr_amd import _lib
from torchdr_amd.affinity import EntropicAffinity
from torchdr_amd.neighbor_embedding.base import NeighborEmbedding, build_transposed_graph
from torchdr_amd.utils.radam import PoincareAdamKernel, RiemannianAdam


class COSNE(NeighborEmbedding):
    r"""Hyperbolic SNE: entropic input affinity, Cauchy kernel :math:`Q_{ij} = \gamma / (d_H(z_i, z_j)^2 + \gamma^2)`
    in the Poincare ball, loss :math:`-\sum P_{ij}\log Q_{ij} + \log\sum_{ij} Q_{ij} + \lambda_1\,
    \mathrm{mean}_i(\|x_i\|^2 - d_H(z_i, 0)^2)^2`, optimised with Riemannian Adam (reference ``cosne.py:162-193``,
    ``utils/radam.py``).  As in the reference the embedding is **float64** (``affinity_matcher.py:552-565``).

    The reference differentiates the loss with autograd over a dense N x N matrix; here the closed-form gradient is
    evaluated by ``tdr_cosne_pairs_f64`` (all pairs streamed through LDS, nothing of size N^2 in memory) and
    ``tdr_cosne_grad_f64`` (both ends of every kNN edge, pull-style through the transposed graph), and the optimizer
    step is ``tdr_radam_poincare_f64``.  With several ranks each rank owns its row chunk: the partition sum is
    all-reduced, the rows are stepped locally and all-gathered (the reference evaluates the whole N x N sum on every
    rank and all-reduces the gradient)."""

    def __init__(self, perplexity: float = 30, learning_rate_for_h_loss: float = 1, gamma: float = 2,
                 n_components: int = 2, lr: Union[float, str] = "auto",
                 optimizer_kwargs: Optional[Union[Dict, str]] = None,
                 scheduler: Optional[Union[str, Type[torch.optim.lr_scheduler.LRScheduler]]] = None,
                 scheduler_kwargs: Optional[Dict] = None, init: str = "hyperbolic", init_scaling: float = 0.5,
                 min_grad_norm: float = 1e-7, max_iter: int = 2000, device: str = "auto", backend=None,
                 verbose: bool = False, random_state: Optional[float] = None, max_iter_affinity: int = 100,
                 metric: str = "sqeuclidean", sparsity: bool = True, check_interval: int = 50, compile: bool = False,
                 distributed: Union[bool, str] = "auto", **kwargs):
        self.metric = metric
        self.perplexity = perplexity
        self.learning_rate_for_h_loss = learning_rate_for_h_loss
        self.gamma = gamma
        self.max_iter_affinity = max_iter_affinity
        self.sparsity = sparsity
        affinity_in = EntropicAffinity(perplexity=perplexity, metric=metric, max_iter=max_iter_affinity,
                                       device=device, backend=backend, verbose=verbose, sparsity=sparsity,
                                       distributed=distributed)
        super().__init__(affinity_in=affinity_in, affinity_out=None, n_components=n_components,
                         optimizer=RiemannianAdam, optimizer_kwargs=optimizer_kwargs, min_grad_norm=min_grad_norm,
                         max_iter=max_iter, lr=lr, scheduler=scheduler, scheduler_kwargs=scheduler_kwargs, init=init,
                         init_scaling=init_scaling, device=device, backend=backend, verbose=verbose,
                         random_state=random_state, check_interval=check_interval, compile=compile,
                         distributed=distributed, **kwargs)

    # --- fit ------------------------------------------------------------------------------------
    def _fit_transform(self, X: torch.Tensor, y: Optional[Any] = None) -> torch.Tensor:
        if not (2 <= self.n_components <= 8):
            raise NotImplementedError("[torchdr_amd] COSNE supports n_components in 2..8.")
        if not self.sparsity and self.world_size > 1:
            raise NotImplementedError("[torchdr_amd] COSNE(sparsity=False) runs in a single process (the dense (N, N) affinity is not sharded).")
        self._x_sqnorm_full = (X.float() ** 2).sum(-1)           # cosne.py:158 (whole set; sliced to the chunk below)
        return super()._fit_transform(X, y)

    def on_affinity_computation_end(self):
        super().on_affinity_computation_end()
        dev = self.affinity_in_.device
        c0 = self.chunk_start_
        self._x_sqnorm = self._x_sqnorm_full.to(dev)[c0:c0 + self.chunk_size_].contiguous()
        del self._x_sqnorm_full
        # sparsity=False: the dense (N, N) affinity as a graph of width N whose row i lists 0 .. N - 1 (`_nn_table` of the base class;
        # the reference's attraction then runs over all pairs, cosne.py:162-171 with NN_indices_ = None)
        self._nn_graph = self.NN_indices_ if self.NN_indices_ is not None else self._nn_table
        self._tgraph = build_transposed_graph(self.affinity_in_, self._nn_graph, c0, self.n_samples_in_,
                                              self.world_size)

    def _init_embedding(self, X):
        """``init='hyperbolic'`` (reference affinity_matcher.py:552-565): expmap0(init_scaling * N(0, 1)) in float64;
        a user array is taken as points of the ball."""
        n = X.shape[0]
        if isinstance(self.init, str):
            if self.init != "hyperbolic":
                raise ValueError(f"[TorchDR] ERROR : init {self.init} not supported in {self.__class__.__name__}.")
            # drawn from the HOST generator (16 bytes per point, once): ``random_state`` then reproduces the initial
            # embedding of the reference's CPU backend, whose trajectories are very sensitive to it
            u = self.init_scaling * torch.randn((n, self.n_components), dtype=torch.float64).to(self.device_)
            un = u.norm(dim=-1, keepdim=True).clamp_min(1e-15)
            emb = un.clamp(-15, 15).tanh() * u / un
        else:
            from torchdr_amd.utils import to_torch

            emb = to_torch(self.init).to(device=self.device_, dtype=torch.float64)
            if emb.shape != (n, self.n_components) or bool(((emb ** 2).sum(-1) >= 1).any()):
                raise ValueError("[torchdr_amd] COSNE: init must be (n_samples, n_components) points inside the unit ball.")
        if self.world_size > 1:
            from torchdr_amd.parallel import broadcast_

            broadcast_(emb)
        self.embedding_ = emb.contiguous()
        return self.embedding_

    # --- optimizer: state lives with the rows this rank owns -----------------------------------------
    def _configure_optimizer(self):
        kw = dict(self.optimizer_kwargs or {})
        self._fused_sgd = False
        self._momentum_buf = None
        self._radam = PoincareAdamKernel(lr=float(self.lr_), **kw)
        self.optimizer_ = self._radam
        return self.optimizer_

    # --- one step -------------------------------------------------------------------------------
    def _euclidean_gradient(self):
        """Closed-form gradient of the reference loss for this rank's rows (float64, (chunk, n_components))."""
        L = _lib.lib()
        n, nc, c0, m = self.n_samples_in_, self.n_components, self.chunk_start_, self.chunk_size_
        st = _lib.stream_ptr()
        Z = self.embedding_
        if getattr(self, "_ws", None) is None:
            self._ws_bytes = int(L.tdr_cosne_workspace_bytes(n, m, nc))
            self._ws = torch.empty(self._ws_bytes // 8 + 1, dtype=torch.float64, device=Z.device)
            self._rowsum = torch.empty(m, dtype=torch.float64, device=Z.device)
            self._egrad = torch.empty((m, nc), dtype=torch.float64, device=Z.device)
        _lib.check(L.tdr_cosne_pairs_f64(_lib.ptr(Z), nc, n, c0, m, float(self.gamma), _lib.ptr(self._rowsum),
                                         _lib.ptr(self._ws), self._ws_bytes, st), "tdr_cosne_pairs_f64")
        S = self._rowsum.sum().reshape(1)
        if self.world_size > 1:
            from torchdr_amd.parallel import allreduce_

            allreduce_(S)
        P = self.affinity_in_
        nn = self.__dict__.get("_nn_graph")
        if nn is None:
            nn = self.NN_indices_
        _lib.check(
            L.tdr_cosne_grad_f64(_lib.ptr(Z), nc, n, c0, m, _lib.ptr(nn), _lib.ptr(P), P.shape[1],
                                 _lib.ptr(self._tgraph[0]), _lib.ptr(self._tgraph[1]), _lib.ptr(self._tgraph[2]),
                                 _lib.ptr(S), _lib.ptr(self._x_sqnorm), float(self.gamma),
                                 float(self.learning_rate_for_h_loss), float(self.early_exaggeration_coeff_),
                                 float(self.repulsion_strength), _lib.ptr(self._ws), self._ws_bytes,
                                 _lib.ptr(self._egrad), st),
            "tdr_cosne_grad_f64",
        )
        return self._egrad

    def _training_step(self):
        egrad = self._euclidean_gradient()
        c0, m = self.chunk_start_, self.chunk_size_
        rows = self.embedding_[c0:c0 + m]
        # like the reference, what is left in ``.grad`` (and enters the convergence check) is the Riemannian gradient
        self._last_grad = self._radam.step(rows, egrad, lr=self._current_lr(), nan_flag=self._nan_flag,
                                           n_iter=int(self.n_iter_))
        self._last_grad_is_chunk = self.world_size > 1
        if self.world_size > 1:
            from torchdr_amd.parallel import allgather_rows_

            allgather_rows_(self.embedding_, c0, m, self.world_size)
        self._lr_pos += 1
        return None

    def clear_memory(self):
        super().clear_memory()
        for attr in ("_tgraph", "_nn_graph", "_ws", "_rowsum", "_egrad", "_x_sqnorm", "_radam"):
            if hasattr(self, attr):
                delattr(self, attr)
