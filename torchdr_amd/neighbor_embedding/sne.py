"""SNE on MI355X -- mirror of ``torchdr/neighbor_embedding/sne.py`` (reference :94-179)."""

from typing import Dict, Optional, Type, Union

import torch

from torchdr_amd import _lib
from torchdr_amd.affinity import EntropicAffinity
from torchdr_amd.neighbor_embedding.base import NeighborEmbedding, build_transposed_graph


class SNE(NeighborEmbedding):
    r"""Stochastic Neighbor Embedding: sparse entropic attraction :math:`\sum_{ij} P_{ij} d_{ij}` on the
    kNN graph and the dense row-wise repulsion :math:`\tfrac1N\sum_i \log\sum_j e^{-d_{ij}}` (diagonal
    included, reference ``sne.py:160-179``).  Closed-form gradient: attraction by ``tdr_ne_grad_f32``
    (kind 2, edge weight :math:`2P_{ij}`), repulsion
    :math:`-\tfrac2N\sum_j e^{-d_{ij}}(1/R_i + 1/R_j)(z_i - z_j)` by two tiled all-pairs passes
    (``tdr_sne_rowsum_f32`` for :math:`R_i=\sum_j e^{-d_{ij}}`, then ``tdr_sne_repulsion_f32``).  With
    several ranks each rank evaluates its row chunk of both passes and the row sums are all-gathered
    (the reference evaluates the whole N x N sum on every rank and divides by the world size,
    ``sne.py:177-178``)."""

    _float64_loop = True   # float64 inputs are embedded in float64 (csrc/tdr_embed_f64.hip: tdr_sne_rowsum_f64 / _repulsion_f64)

    def __init__(self, perplexity: float = 30, n_components: int = 2, lr: Union[float, str] = "auto",
                 optimizer: Union[str, Type[torch.optim.Optimizer]] = "SGD",
                 optimizer_kwargs: Union[Dict, str] = "auto",
                 scheduler: Optional[Union[str, Type[torch.optim.lr_scheduler.LRScheduler]]] = None,
                 scheduler_kwargs: Optional[Dict] = None, init: str = "pca", init_scaling: float = 1e-4,
                 min_grad_norm: float = 1e-7, max_iter: int = 2000, device: str = "auto", backend=None,
                 verbose: bool = False, random_state: Optional[float] = None, max_iter_affinity: int = 100,
                 metric: str = "sqeuclidean", sparsity: bool = True,
                 early_exaggeration_coeff: Optional[float] = None, early_exaggeration_iter: Optional[int] = None,
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

    def _float64_ok(self, X) -> bool:
        return int(self.n_components) <= 16      # register instances of the float64 all-pairs kernels

    def _compute_gradients(self):
        L = _lib.lib()
        n, nc = self.n_samples_in_, self.n_components
        st = _lib.stream_ptr()
        P = self.affinity_in_
        dt = P.dtype     # float64 inputs are embedded in float64 (csrc/tdr_embed_f64.hip), like the reference
        grad = torch.zeros((n, nc), dtype=dt, device=self.device_)
        _lib.check(
            _lib.fn("tdr_ne_grad", dt)(_lib.ptr(self.embedding_), nc, n, self.chunk_start_, self.chunk_size_,
                                       _lib.ptr(self._nn_table), _lib.ptr(P), P.shape[1], _lib.ptr(self._tgraph[0]),
                                       _lib.ptr(self._tgraph[1]), _lib.ptr(self._tgraph[2]), 2,
                                       float(self.early_exaggeration_coeff_), 0.0, 0, None, 0, int(self.n_iter_),
                                       _lib.ptr(grad), st),
            "tdr_ne_grad",
        )
        R = torch.empty((self.chunk_size_, 1), dtype=dt, device=self.device_)
        _lib.check(
            _lib.fn("tdr_sne_rowsum", dt)(_lib.ptr(self.embedding_), nc, n, self.chunk_start_, self.chunk_size_, _lib.ptr(R), st),
            "tdr_sne_rowsum",
        )
        if self.world_size > 1:
            from torchdr_amd.parallel import allgather_rows

            R = allgather_rows(R, n, self.world_size)
        _lib.check(
            _lib.fn("tdr_sne_repulsion", dt)(_lib.ptr(self.embedding_), nc, n, self.chunk_start_, self.chunk_size_,
                                             _lib.ptr(R), -2.0 * float(self.repulsion_strength) / n, _lib.ptr(grad), st),
            "tdr_sne_repulsion",
        )
        return grad, False
