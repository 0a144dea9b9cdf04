"""t-SNE on MI355X -- mirror of ``torchdr/neighbor_embedding/tsne.py`` (reference :94-180)."""

from typing import Dict, Optional, Type, Union

import torch

from torchdr_amd import _lib
from torchdr_amd.affinity import EntropicAffinity
from torchdr_amd.neighbor_embedding.base import NeighborEmbedding, build_transposed_graph


class TSNE(NeighborEmbedding):
    r"""t-SNE: sparse entropic attraction :math:`\sum_{ij} P_{ij}\log(1 + d_{ij})` on the kNN graph and
    the exact dense repulsion :math:`\log\sum_{ij}(1 + d_{ij})^{-1}` (diagonal included, reference
    ``tsne.py:162-180``).  Gradients in closed form: attraction by ``tdr_ne_grad_f32`` (kind 1),
    repulsion :math:`-(4/S)\sum_j (z_i - z_j)/(1 + d_{ij})^2` by the tiled all-pairs kernel
    ``tdr_tsne_repulsion_f32``; with several ranks each rank evaluates only its row chunk of the
    N x N sum and the partition function S is all-reduced (the reference recomputes the full sum on
    every rank, ``tsne.py:178-179``).  Early exaggeration 12 for 250 iterations, ``lr="auto"``,
    SGD momentum 0.5 -> 0.8 with the optimizer rebuilt at the switch."""

    _relabel_rect = True   # single GPU, pruned search: the loop runs in the kNN stage's cluster-sorted numbering
    _float64_loop = True   # float64 inputs are embedded in float64 (csrc/tdr_embed_f64.hip; n_components <= 16)

    def _float64_ok(self, X) -> bool:
        # tdr_tsne_repulsion_f64 has register instances up to 16 components; the float32 kernels go to 32
        return int(self.n_components) <= 16

    def __init__(self, perplexity: float = 30, n_components: int = 2, lr: Union[float, str] = "auto",
                 optimizer: Union[str, Type[torch.optim.Optimizer]] = "SGD",
                 optimizer_kwargs: Union[Dict, str] = "auto",
                 scheduler: Optional[Union[str, Type[torch.optim.lr_scheduler.LRScheduler]]] = None,
                 scheduler_kwargs: Optional[Dict] = None, init: str = "pca", init_scaling: float = 1e-4,
                 min_grad_norm: float = 1e-7, max_iter: int = 2000, device: str = "auto", backend=None,
                 verbose: bool = False, random_state: Optional[float] = None,
                 early_exaggeration_coeff: float = 12.0, early_exaggeration_iter: int = 250,
                 max_iter_affinity: int = 100, metric: str = "sqeuclidean", sparsity: bool = True,
                 check_interval: int = 50, compile: bool = False, distributed: Union[bool, str] = "auto",
                 **kwargs):
        self.metric = metric
        self.perplexity = perplexity
        self.max_iter_affinity = max_iter_affinity
        self.sparsity = sparsity
        affinity_in = EntropicAffinity(perplexity=perplexity, metric=metric, max_iter=max_iter_affinity,
                                       device=device, backend=backend, verbose=verbose, sparsity=sparsity,
                                       distributed=distributed)
        super().__init__(affinity_in=affinity_in, n_components=n_components, optimizer=optimizer,
                         optimizer_kwargs=optimizer_kwargs, min_grad_norm=min_grad_norm, max_iter=max_iter, lr=lr,
                         scheduler=scheduler, scheduler_kwargs=scheduler_kwargs, init=init,
                         init_scaling=init_scaling, device=device, backend=backend, verbose=verbose,
                         random_state=random_state, early_exaggeration_coeff=early_exaggeration_coeff,
                         early_exaggeration_iter=early_exaggeration_iter, check_interval=check_interval,
                         compile=compile, distributed=distributed, **kwargs)

    def on_affinity_computation_end(self):
        super().on_affinity_computation_end()
        self._tgraph = build_transposed_graph(self.affinity_in_, self._nn_table, self.chunk_start_,
                                              self.n_samples_in_, self.world_size)

    def clear_memory(self):
        super().clear_memory()
        if hasattr(self, "_tgraph"):
            delattr(self, "_tgraph")

    def _compute_gradients(self):
        L = _lib.lib()
        n, nc = self.n_samples_in_, self.n_components
        st = _lib.stream_ptr()
        P = self.affinity_in_
        dt = P.dtype
        grad = torch.zeros((n, nc), dtype=dt, device=self.device_)
        _lib.check(
            _lib.fn("tdr_ne_grad", dt)(_lib.ptr(self.embedding_), nc, n, self.chunk_start_, self.chunk_size_,
                              _lib.ptr(self._nn_table), _lib.ptr(P), P.shape[1], _lib.ptr(self._tgraph[0]),
                              _lib.ptr(self._tgraph[1]), _lib.ptr(self._tgraph[2]), 1,
                              float(self.early_exaggeration_coeff_), 0.0, 0, None, 0, int(self.n_iter_),
                              _lib.ptr(grad), st),
            "tdr_ne_grad",
        )
        F = torch.empty((self.chunk_size_, nc), dtype=dt, device=self.device_)
        S = torch.zeros(1, dtype=torch.float64, device=self.device_)
        ws_bytes = int(L.tdr_tsne_repulsion_workspace_bytes(n, self.chunk_size_, nc)) if dt == torch.float32 else 0
        if ws_bytes > 0:    # columns spread over several workgroups per row block (N <= ~1M: a row block alone is too coarse)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device_)
            _lib.check(L.tdr_tsne_repulsion_split_f32(_lib.ptr(self.embedding_), nc, n, self.chunk_start_, self.chunk_size_,
                                                      _lib.ptr(F), _lib.ptr(S), _lib.ptr(ws), ws_bytes, st),
                       "tdr_tsne_repulsion_split_f32")
        else:
            _lib.check(
                _lib.fn("tdr_tsne_repulsion", dt)(_lib.ptr(self.embedding_), nc, n, self.chunk_start_, self.chunk_size_,
                                                  _lib.ptr(F), _lib.ptr(S), st),
                "tdr_tsne_repulsion",
            )
        if self.world_size > 1:
            from torchdr_amd.parallel import allreduce_

            allreduce_(S)
        rows = grad[self.chunk_start_: self.chunk_start_ + self.chunk_size_]
        _lib.check(
            _lib.fn("tdr_add_scaled", dt)(_lib.ptr(rows), _lib.ptr(F), _lib.ptr(S), -4.0 * float(self.repulsion_strength),
                                          rows.numel(), st),
            "tdr_add_scaled",
        )
        return grad, False
