"""LargeVis on MI355X -- mirror of ``torchdr/neighbor_embedding/largevis.py`` (reference :108-201)."""

from typing import Dict, Optional, Type, Union

import torch

from torchdr_amd import _lib
from torchdr_amd.affinity import EntropicAffinity
from torchdr_amd.neighbor_embedding.base import NegativeSamplingNeighborEmbedding, build_transposed_graph


class LargeVis(NegativeSamplingNeighborEmbedding):
    r"""LargeVis: entropic input affinity, :math:`Q_{ij} = 1/(2 + \|z_i - z_j\|^2)`, loss
    :math:`-\sum_{ij} P_{ij}\log Q_{ij} - \tfrac1N\sum_{i,\,j\in\mathrm{Neg}(i)}\log(1 - Q_{ij})`
    (reference ``largevis.py:181-201``).  Its gradient is evaluated in closed form by
    ``tdr_ne_grad_f32`` (both endpoints of every edge move, as with autograd's gather backward).
    Defaults as the reference: ``lr="auto"`` (= max(N/4, 50)), SGD momentum 0.8, ``LinearLR`` with
    torch's default arguments (lr ramps 1/3 -> 1 over the first 5 steps)."""

    _relabel_rect = True   # single GPU, pruned search: the loop runs in the kNN stage's cluster-sorted numbering
    _float64_loop = True   # float64 inputs are embedded in float64 (csrc/tdr_embed_f64.hip)

    def __init__(self, perplexity: float = 30, n_components: int = 2, lr: Union[float, str] = "auto",
                 optimizer: Union[str, Type[torch.optim.Optimizer]] = "SGD",
                 optimizer_kwargs: Union[Dict, str] = "auto",
                 scheduler: Optional[Union[str, Type[torch.optim.lr_scheduler.LRScheduler]]] = "LinearLR",
                 scheduler_kwargs: Optional[Dict] = None, init: str = "pca", init_scaling: float = 1e-4,
                 min_grad_norm: float = 1e-7, max_iter: int = 1000, device: str = "auto", backend="faiss",
                 verbose: bool = False, random_state: Optional[float] = None, max_iter_affinity: int = 100,
                 metric: str = "sqeuclidean", n_negatives: int = 5, sparsity: bool = True,
                 early_exaggeration_coeff: Optional[float] = None, early_exaggeration_iter: Optional[int] = None,
                 check_interval: int = 50, discard_NNs: bool = False, compile: bool = False,
                 distributed: Union[bool, str] = "auto", **kwargs):
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
                         early_exaggeration_iter=early_exaggeration_iter, n_negatives=n_negatives,
                         check_interval=check_interval, discard_NNs=discard_NNs, compile=compile,
                         distributed=distributed, **kwargs)

    def on_affinity_computation_end(self):
        super().on_affinity_computation_end()
        self._tgraph = build_transposed_graph(self.affinity_in_, self._nn_table, self.chunk_start_,
                                              self.n_samples_in_, self.world_size)

    def _compute_gradients(self):
        n, nc = self.n_samples_in_, self.n_components
        from torchdr_amd.neighbor_embedding import base as _nb

        P = self.affinity_in_
        grad = torch.zeros((n, nc), dtype=P.dtype, device=self.device_)
        nn = self._nn_table
        neg = self._neg_ptr_tensor()
        L = _lib.lib()
        if (neg is None and _nb._opt("PERM_NEGATIVES") == "runs" and P.dtype == torch.float32
                and L.tdr_ne_grad_runs_supported(nc, n, int(self.n_negatives)) and self.embedding_.data_ptr() % 16 == 0):
            # no injected table: RUN-permutation sampler -- both shares of every pair pulled by the rows themselves, the negatives
            # staged into LDS run by run (csrc/tdr_embed.hip: ne_pull4_runs_kernel).  The gradient of a rank's rows is COMPLETE
            # (nothing is sent to other rows): in a row-sharded fit the rank hands back its chunk (`rows_only`: it steps its rows
            # and the rows are all-gathered, affinity_matcher.py) -- the same sampler, keyed by global rows, for every world size
            gchunk = grad if self.world_size == 1 else torch.empty((self.chunk_size_, nc), dtype=P.dtype, device=self.device_)
            _lib.check(
                L.tdr_ne_grad_runs_f32(
                    _lib.ptr(self.embedding_), nc, n, self.chunk_start_, self.chunk_size_, _lib.ptr(nn), _lib.ptr(P), P.shape[1],
                    _lib.ptr(self._tgraph[0]), _lib.ptr(self._tgraph[1]), _lib.ptr(self._tgraph[2]),
                    float(self.early_exaggeration_coeff_), float(self.repulsion_strength) * 2.0 / n, int(self.n_negatives),
                    self._neg_seed, int(self.n_iter_), _lib.ptr(gchunk), _lib.stream_ptr(),
                ),
                "tdr_ne_grad_runs_f32",
            )
            return gchunk, self.world_size > 1
        if neg is None and _nb._opt("PERM_NEGATIVES") and self.world_size == 1 and P.dtype == torch.float32 and self.n_negatives > 0:
            # one GPU, no injected table: permutation sampler, every pair's two shares pulled (no atomics)
            _lib.check(
                _lib.lib().tdr_ne_grad_perm_f32(
                    _lib.ptr(self.embedding_), nc, n, 0, n, _lib.ptr(nn), _lib.ptr(P), P.shape[1], _lib.ptr(self._tgraph[0]),
                    _lib.ptr(self._tgraph[1]), _lib.ptr(self._tgraph[2]), 0, float(self.early_exaggeration_coeff_),
                    float(self.repulsion_strength) * 2.0 / n, int(self.n_negatives), self._neg_seed, int(self.n_iter_), None,
                    _lib.ptr(grad), _lib.stream_ptr(),
                ),
                "tdr_ne_grad_perm_f32",
            )
            return grad, False
        _lib.check(
            _lib.fn("tdr_ne_grad", P.dtype)(
                _lib.ptr(self.embedding_), nc, n, self.chunk_start_, self.chunk_size_, _lib.ptr(nn), _lib.ptr(P),
                P.shape[1], _lib.ptr(self._tgraph[0]), _lib.ptr(self._tgraph[1]), _lib.ptr(self._tgraph[2]), 0,
                float(self.early_exaggeration_coeff_), float(self.repulsion_strength) * 2.0 / n,
                int(self.n_negatives), _lib.ptr(neg), self._neg_seed, int(self.n_iter_), _lib.ptr(grad),
                _lib.stream_ptr(),
            ),
            "tdr_ne_grad",
        )
        return grad, False

    def clear_memory(self):
        super().clear_memory()
        if hasattr(self, "_tgraph"):
            delattr(self, "_tgraph")
