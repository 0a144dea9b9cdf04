"""Neighbor-embedding base classes -- mirror of ``torchdr/neighbor_embedding/base.py``
(``NeighborEmbedding`` :20-423, ``NegativeSamplingNeighborEmbedding`` :426-649)."""

import os
import warnings
from typing import Any, Dict, Optional, Type, Union

import numpy as np
import torch
import torch.distributed as dist

from torchdr_amd.affinity import Affinity
from torchdr_amd.affinity_matcher import AffinityMatcher


# per-iteration exchange through an RCCL communicator owned by the C library (tdr_ctx_*: collectives enqueued on the
# compute stream, graph-capturable) when the process group runs on RCCL; False keeps every collective in torch.distributed
RCCL_CONTEXT = True
# LargeVis / InfoTSNE on one GPU: negatives drawn as keyed permutations of the rows, both shares of a pair pulled by the rows
# themselves (tdr_ne_grad_perm_f32: no far-endpoint atomics; csrc/tdr_embed_common.h).  False = the hash sampler + atomics (the
# reference's independent draws, base.py:628-636; what every row-sharded fit and every injected table uses).  "runs" (LargeVis,
# round 6): the permutation over RUNS of 16 consecutive rows, negatives served from LDS (tdr_ne_grad_runs_f32; InfoTSNE treats it
# as True).  Default since round 6 after the quality gate of tests/test_embed_gpu.py::test_largevis_samplers_reach_the_reference_
# scores_on_four_regimes (profiles/r06_largevis_quality.jsonl): C3's gradient launch 0.211 -> 0.143 ms, fit 146 -> 107 ms.
PERM_NEGATIVES = "runs"
# PEER_EXCHANGE: row-sharded fits exchange the rows they stepped as direct peer writes over xGMI (parallel.PeerExchange,
# csrc/tdr_peerx.hip) instead of an RCCL ring all-gather; falls back to RCCL / torch.distributed when HIP IPC or the stress
# self-check fails on any rank.  True: always try; False: never; "auto": only where ranks SHARE a device (more ranks than GPUs:
# the configurations this build could test -- there it replaces host-staged collectives).  Between distinct devices the path has
# never run: a wrong assumption about peer mappings would be a memory fault, not a failed self-check, so it is opt-in there
# (`bench.py --peer-exchange`, `config.options(PEER_EXCHANGE=True)`) until it has been seen on an 8-GPU node.
PEER_EXCHANGE = "auto"

def _opt(name):
    """A behaviour switch of this module: the scoped override (torchdr_amd.config.options) or the module attribute."""
    from torchdr_amd import config

    return config.get(name, globals())



class NeighborEmbedding(AffinityMatcher):
    _lr_as_tensor = False

    def __init__(self, affinity_in: Affinity, affinity_out: Optional[Affinity] = None,
                 kwargs_affinity_out: Optional[Dict] = None, n_components: int = 2,
                 lr: Union[float, str] = 1e0, optimizer: Union[str, Type[torch.optim.Optimizer]] = "SGD",
                 optimizer_kwargs: Union[Dict, str] = "auto",
                 scheduler: Optional[Union[str, Type[torch.optim.lr_scheduler.LRScheduler]]] = None,
                 scheduler_kwargs: Union[Dict, str, None] = "auto", min_grad_norm: float = 1e-7,
                 max_iter: int = 2000, init: Union[str, torch.Tensor, np.ndarray] = "pca",
                 init_scaling: float = 1e-4, device: str = "auto", backend=None, verbose: bool = False,
                 random_state: Optional[float] = None, early_exaggeration_coeff: Optional[float] = None,
                 early_exaggeration_iter: Optional[int] = None, repulsion_strength: float = 1.0,
                 check_interval: int = 50, compile: bool = False, distributed: Union[bool, str] = "auto",
                 **kwargs: Any):
        self.early_exaggeration_iter = early_exaggeration_iter if early_exaggeration_iter is not None else 0
        self.early_exaggeration_coeff = early_exaggeration_coeff if early_exaggeration_coeff is not None else 1
        self.repulsion_strength = repulsion_strength
        # extension of the reference's surface: fit_transform receives this rank's ROW SHARD (chunk rule of
        # DistributedContext) instead of the full block on every rank; the shards are all-gathered over RCCL first
        self.sharded_input = bool(kwargs.pop("sharded_input", False))
        if "learning_rate" in kwargs:  # sklearn-style aliases (reference :170-173)
            lr = kwargs.pop("learning_rate")
        if "early_exaggeration" in kwargs:
            self.early_exaggeration_coeff = kwargs.pop("early_exaggeration")
        _scheduler_kwargs = scheduler_kwargs
        if scheduler == "LinearLR" and isinstance(scheduler_kwargs, str) and scheduler_kwargs == "auto":
            # by default the linear schedule goes from 1 to 0 over max_iter (reference :176-182)
            _scheduler_kwargs = {"start_factor": torch.tensor(1.0), "end_factor": torch.tensor(0),
                                 "total_iters": max_iter}
        elif isinstance(scheduler_kwargs, str) and scheduler_kwargs == "auto":
            _scheduler_kwargs = None
        super().__init__(affinity_in=affinity_in, affinity_out=affinity_out,
                         kwargs_affinity_out=kwargs_affinity_out, n_components=n_components, optimizer=optimizer,
                         optimizer_kwargs=optimizer_kwargs, lr=lr, scheduler=scheduler,
                         scheduler_kwargs=_scheduler_kwargs, min_grad_norm=min_grad_norm, max_iter=max_iter,
                         init=init, init_scaling=init_scaling, device=device, backend=backend, verbose=verbose,
                         random_state=random_state, check_interval=check_interval, compile=compile, **kwargs)
        self._setup_distributed(distributed)
        self._perm = None
        self.loop_order_ = None     # after a fit: caller's row of every loop row, or None when the loop ran in the caller's numbering

    # --- fit ------------------------------------------------------------------------------------
    def _check_n_neighbors(self, n):
        for name in ("perplexity", "n_neighbors"):
            if hasattr(self, name):
                value = getattr(self, name)
                if n <= value:
                    raise ValueError(
                        f"[TorchDR] ERROR : Number of samples is smaller than {name} ({n} <= {value})."
                    )
        return self

    def _fit_transform(self, X: torch.Tensor, y: Optional[Any] = None) -> torch.Tensor:
        self._check_n_neighbors(X.shape[0])
        self.early_exaggeration_coeff_ = self.early_exaggeration_coeff
        self._perm = None
        self.loop_order_ = None     # kept after the fit: caller's row of every loop row, or None when the loop ran unrelabelled
        Z = super()._fit_transform(X, y)
        perm = getattr(self, "_perm", None)
        if perm is not None:    # the loop ran in the kNN stage's cluster-sorted numbering: back to the caller's row order
            out = torch.empty_like(Z)
            out.index_copy_(0, perm, Z)
            self.embedding_ = Z = out
            self._perm = None
        return Z

    # ---- loop numbering -------------------------------------------------------------------------------------------------
    # The pruned kNN search works in a cluster-sorted row order (ClusterIndex.perm, deterministic).  When nothing outside
    # the shipped classes looks at rows during the optimisation, the loop numbers the points in that order: a row's
    # neighbours then sit in the same few cache lines of the embedding and the gathers of the neighbour edges hit the
    # L1 / L2 instead of the fabric (UMAP: 0.300 -> 0.274 ms per iteration at N = 1M; LargeVis at N = 1M, kNN width 15:
    # see DESIGN.md).  The embedding is permuted on the way in and un-permuted on the way out; the negative sampler is
    # keyed by loop row numbers (same distribution, another stream).  UMAP renumbers its CSR graph itself
    # (`UMAP._relabel`); estimators on the rectangular (n, k) graph opt in with `_relabel_rect`.
    _relabel_rect = False

    def _rect_relabel_eligible(self) -> bool:
        from torchdr_amd.neighbor_embedding import umap as _umap

        if not (_umap._opt("RELABEL") and self._relabel_rect) or self.world_size > 1:
            return False
        if getattr(self, "discard_NNs", False) or getattr(self, "neg_indices_", None) is not None:
            return False
        # a subclass defined outside the package may look at rows in its hooks: it keeps the caller's numbering
        return type(self).__module__.startswith("torchdr_amd.")

    def _relabel_rect_graph(self):
        if not self._relabel_rect:      # UMAP reads the order itself (CSR graph)
            return
        order = getattr(self.affinity_in, "_row_order", None)
        if isinstance(self.affinity_in, Affinity):
            self.affinity_in._row_order = None
        P, NN = self._buffers.get("affinity_in_"), self._buffers.get("NN_indices_")
        if order is None or not self._rect_relabel_eligible() or NN is None or not torch.is_tensor(P):
            return
        perm, inv = order
        n = self.n_samples_in_
        if perm.numel() != n or P.shape[0] != n or NN.shape[0] != n or NN.shape[1] == n:
            return
        p64 = perm.to(torch.int64)
        self._buffers["affinity_in_"] = P.index_select(0, p64).contiguous()
        self._buffers["NN_indices_"] = inv[NN.index_select(0, p64).long()].to(NN.dtype).contiguous()
        self._perm = self.loop_order_ = p64

    # --- loss hooks of the autograd mode (reference :207-231); the estimators of this package override
    #     _compute_gradients with closed forms and never evaluate these -------------------------------
    def _compute_attractive_loss(self):
        raise NotImplementedError("[TorchDR] ERROR : _compute_attractive_loss method must be implemented.")

    def _compute_repulsive_loss(self):
        raise NotImplementedError("[TorchDR] ERROR : _compute_repulsive_loss method must be implemented.")

    def _compute_loss(self):
        return (self.early_exaggeration_coeff_ * self._compute_attractive_loss()
                + self.repulsion_strength * self._compute_repulsive_loss())

    # closed-form mode of the reference (:236-254): a subclass with `_use_closed_form_gradients = True` supplies the two
    # gradient terms (rows of its chunk) as torch tensors; the estimators of this package override `_compute_gradients`
    # itself with their HIP kernels
    def _compute_gradients(self):
        return (self.early_exaggeration_coeff_ * self._compute_attractive_gradients()
                + self.repulsion_strength * self._compute_repulsive_gradients())

    _compute_gradients._is_base = True

    def _compute_attractive_gradients(self):
        raise NotImplementedError(
            "[TorchDR] ERROR : _compute_attractive_gradients method must be implemented "
            "when _use_closed_form_gradients is True."
        )

    def _compute_repulsive_gradients(self):
        raise NotImplementedError(
            "[TorchDR] ERROR : _compute_repulsive_gradients method must be implemented "
            "when _use_closed_form_gradients is True."
        )

    # --- early exaggeration (reference :282-295) ------------------------------------------------
    def on_training_step_end(self):
        if self.early_exaggeration_coeff_ > 1 and int(self.n_iter_) == self.early_exaggeration_iter:
            self.early_exaggeration_coeff_ = 1
            self._set_learning_rate()
            self._configure_optimizer()  # rebuilds the optimizer: momentum buffer reset, new momentum
            self._configure_scheduler()
        return self

    # --- auto learning rate / optimizer (reference :299-350) ------------------------------------
    def _set_learning_rate(self):
        if isinstance(self.lr, str) and self.lr == "auto":
            if self.optimizer != "SGD" and self.verbose:
                warnings.warn("[TorchDR] WARNING : when 'auto' is used for the learning rate, the optimizer "
                              "should be 'SGD'.")
            self.lr_ = max(self.n_samples_in_ / self.early_exaggeration_coeff_ / 4, 50)
        else:
            self.lr_ = self.lr

    def _resolve_optimizer_kwargs(self):
        if isinstance(self.optimizer_kwargs, str) and self.optimizer_kwargs == "auto":
            if self.optimizer == "SGD":
                return {"momentum": 0.5} if self.early_exaggeration_coeff_ > 1 else {"momentum": 0.8}
            return {}
        return self.optimizer_kwargs or {}

    def _configure_scheduler(self, n_iter=None):
        if self.early_exaggeration_coeff_ > 1:
            n_iter = min(self.early_exaggeration_iter, self.max_iter)
        else:
            n_iter = self.max_iter - self.early_exaggeration_iter
        return super()._configure_scheduler(n_iter)

    # --- distributed (reference :354-423) -------------------------------------------------------
    def _setup_distributed(self, distributed):
        if isinstance(distributed, str) and distributed == "auto":
            self.distributed = dist.is_available() and dist.is_initialized()
        else:
            self.distributed = bool(distributed)
        if self.distributed:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError(
                    "[TorchDR] distributed=True requires launching with torchrun. "
                    "Example: torchrun --nproc_per_node=4 your_script.py"
                )
            self.rank = dist.get_rank()
            self.world_size = dist.get_world_size()
            self.is_multi_gpu = self.world_size > 1
            local_rank = int(os.environ.get("LOCAL_RANK", 0))
            if torch.cuda.is_available():
                local_rank %= torch.cuda.device_count()  # more ranks than GPUs only in the gloo debugging mode
                torch.cuda.set_device(local_rank)
            if self.device == "cpu":
                raise ValueError("[TorchDR] Distributed mode requires GPU (device cannot be 'cpu')")
            self.device = torch.device(f"cuda:{local_rank}")
        else:
            self.rank = 0
            self.world_size = 1
            self.is_multi_gpu = False

    def on_affinity_computation_end(self):
        super().on_affinity_computation_end()
        if hasattr(self.affinity_in, "chunk_start_"):
            self.chunk_start_ = self.affinity_in.chunk_start_
            self.chunk_size_ = self.affinity_in.chunk_size_
        elif self.world_size > 1:
            raise ValueError(
                "[TorchDR] ERROR: Distributed mode is enabled but affinity_in does not have chunk bounds. "
                "Make sure affinity_in has distributed=True."
            )
        else:
            self.chunk_start_ = 0
            self.chunk_size_ = self.n_samples_in_
        # Dense input affinity (sparsity=False): NN_indices_ is None as in the reference, whose attraction then runs over
        # all pairs (pairwise_distances_indexed with key_indices=None, e.g. tsne.py:162-170).  The edge kernels see the
        # same thing as a rectangular graph of width N whose row i lists 0..N-1.
        # (read from the buffer dict: UMAP exposes both names as lazily materialised properties)
        if self.world_size == 1:
            self._relabel_rect_graph()
        self._nn_table = self._buffers.get("NN_indices_")
        P = self._buffers.get("affinity_in_")
        if self._nn_table is None and torch.is_tensor(P) and P.dim() == 2 and P.shape[1] == self.n_samples_in_:
            cols = torch.arange(self.n_samples_in_, dtype=torch.int32, device=P.device)
            self._nn_table = cols.unsqueeze(0).expand(P.shape[0], -1).contiguous()
        self._rccl_ctx = None
        px = _opt("PEER_EXCHANGE")
        if px == "auto":
            px = torch.cuda.is_available() and self.world_size > torch.cuda.device_count()
        # only estimators whose step exchanges ROWS use a context (closed-form gradients of the chunk's rows: UMAP and user
        # subclasses following the reference's contract, affinity_matcher.py:384-413); LargeVis / TSNE / SNE all-reduce a full
        # gradient and COSNE gathers through torch.distributed -- they would pay the IPC mapping, the stress self-check and the
        # staging memory for nothing
        uses_rows = bool(getattr(self, "_use_closed_form_gradients", False))
        if not uses_rows:
            px = False
        if self.world_size > 1 and px and torch.cuda.is_available() and getattr(self, "_dtype", torch.float32) == torch.float32:
            # the rows every rank stepped travel as direct peer writes (csrc/tdr_peerx.hip); None when the peers cannot be
            # mapped or the stress self-check fails on any rank
            from torchdr_amd.parallel import PeerExchange

            self._rccl_ctx = PeerExchange.shared(self.n_samples_in_, self.n_components, self.device_)
        if self._rccl_ctx is None and uses_rows and self.world_size > 1 and _opt("RCCL_CONTEXT") and dist.get_backend() == "nccl" and torch.cuda.is_available():
            from torchdr_amd.parallel import RcclContext

            self._rccl_ctx = RcclContext.shared(self.n_samples_in_, self.device_)   # one communicator per process
        from torchdr_amd import parallel as _par

        if _par.EMULATION is not None and uses_rows and self.world_size > 1:    # one rank of W run alone: loopback exchange
            self._rccl_ctx = _par.EMULATION.exchange(self.n_samples_in_, self.n_components, self.device_)
        # how a row-sharded fit exchanged its rows (kept after clear_memory): "PeerExchange", "RcclContext" or "torch.distributed"
        self.row_exchange_ = (type(self._rccl_ctx).__name__ if self._rccl_ctx is not None else "torch.distributed") if self.world_size > 1 else None

    def clear_memory(self):
        super().clear_memory()
        self._nn_table = None
        self._rccl_ctx = None    # the communicator itself is the process's (RcclContext.shared): not destroyed per fit

    @property
    def chunk_indices_(self):
        return torch.arange(self.chunk_start_, self.chunk_start_ + self.chunk_size_, device=self.device_)

    def _init_embedding(self, X: torch.Tensor):
        emb = super()._init_embedding(X)
        if getattr(self, "_perm", None) is not None:   # rows of the initial embedding in the loop's numbering
            self.embedding_ = emb.index_select(0, self._perm).contiguous()
        from torchdr_amd.affinity_matcher import pca_scores

        same_on_every_rank = isinstance(self.init, str) and self.init == "pca" and getattr(self, "_pca_deterministic", pca_scores.deterministic)
        if self.world_size > 1 and not same_on_every_rank:
            # reference :421.  init="pca" needs no exchange WHEN the deterministic PCA kernels produced it (float32 block on
            # the device, D <= 256, <= 4 components, PCA_EIGH = "jacobi": ordered fp64 combination of the Gram tiles, one-
            # workgroup Jacobi -- same bits on every rank); the torch fallback (library GEMM / eigh: split-K or atomic
            # kernels may differ in rounding between ranks) keeps the reference's broadcast
            from torchdr_amd.parallel import broadcast_
            from torchdr_amd.utils.phases import phase

            with phase("init: broadcast"):
                broadcast_(self.embedding_, src=0)
        return self.embedding_


def build_transposed_graph(P, NN, chunk_start, n_total, world_size):
    """In-edges of this rank's rows of the rectangular (n, k) graph as CSR (rowptr int64, src int32 global,
    val fp32): edge i -> j with weight P_ij becomes an entry of row j.  Lets the gradient kernel evaluate the
    far-endpoint share of every neighbour edge pull-style (no atomics).  Edges whose target row lives on another
    rank are routed there with the same all-to-all-v as the symmetrisation."""
    n, k = P.shape
    dev = P.device
    src = (torch.arange(n, device=dev, dtype=torch.int64) + chunk_start).repeat_interleave(k)
    dst = NN.reshape(-1).to(torch.int64)
    val = P.reshape(-1)
    if world_size > 1:
        from torchdr_amd.parallel import exchange_transposed_edges

        local = (dst >= chunk_start) & (dst < chunk_start + n)
        er, ec, ev = exchange_transposed_edges(P, NN, chunk_start, n_total, world_size)
        rows = torch.cat([dst[local] - chunk_start, er.to(torch.int64)])
        srcs = torch.cat([src[local], ec.to(torch.int64)])
        vals = torch.cat([val[local], ev])
    else:
        rows, srcs, vals = dst, src, val
    # in-edges of a row by ascending source row: the order of a row's sum then does not depend on how the edges arrived (local
    # edges first, then the exchanged ones rank by rank) -- a row-sharded fit adds the same terms in the same order as one process
    order = torch.argsort(rows * int(n_total) + srcs, stable=True)
    counts = torch.bincount(rows, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = counts.cumsum(0)
    return rowptr, srcs[order].to(torch.int32).contiguous(), vals[order].contiguous()


class NegativeSamplingNeighborEmbedding(NeighborEmbedding):
    """Repulsion through per-step negative samples (reference :426-649).  Negatives are drawn
    in-kernel (counter hash keyed by seed / iteration / row / column) uniformly from all points except
    the row itself (``randint(0, N-1)`` then ``+1 if >= self``, reference :628-636); setting
    ``neg_indices_`` (tests) injects an explicit table instead.  ``discard_NNs=True`` uses the
    reference's exclusion-table scheme (:639-647) evaluated with torch ops and injected."""

    def __init__(self, affinity_in: Affinity, affinity_out: Optional[Affinity] = None,
                 kwargs_affinity_out: Optional[Dict] = None, n_components: int = 2,
                 lr: Union[float, str] = 1e0, optimizer="SGD", optimizer_kwargs: Union[Dict, str] = "auto",
                 scheduler=None, scheduler_kwargs: Union[Dict, str, None] = "auto", min_grad_norm: float = 1e-7,
                 max_iter: int = 2000, init="pca", init_scaling: float = 1e-4, device: str = "auto", backend=None,
                 verbose: bool = False, random_state: Optional[float] = None,
                 early_exaggeration_coeff: float = 1.0, early_exaggeration_iter: Optional[int] = None,
                 repulsion_strength: float = 1.0, n_negatives: int = 5, check_interval: int = 50,
                 discard_NNs: bool = False, compile: bool = False, **kwargs):
        super().__init__(affinity_in=affinity_in, affinity_out=affinity_out,
                         kwargs_affinity_out=kwargs_affinity_out, n_components=n_components, lr=lr,
                         optimizer=optimizer, optimizer_kwargs=optimizer_kwargs, scheduler=scheduler,
                         scheduler_kwargs=scheduler_kwargs, min_grad_norm=min_grad_norm, max_iter=max_iter,
                         init=init, init_scaling=init_scaling, device=device, backend=backend, verbose=verbose,
                         random_state=random_state, early_exaggeration_coeff=early_exaggeration_coeff,
                         early_exaggeration_iter=early_exaggeration_iter, repulsion_strength=repulsion_strength,
                         check_interval=check_interval, compile=compile, **kwargs)
        self.n_negatives = n_negatives
        self.discard_NNs = discard_NNs
        self.neg_indices_ = None

    def on_affinity_computation_end(self):
        super().on_affinity_computation_end()
        # ONE seed for all ranks (rank 0's): the sampler is keyed by (seed, iteration, GLOBAL row, column), so every row has
        # its own stream whichever rank evaluates it, and a row-sharded fit draws exactly the negatives of the
        # single-process fit with the same random_state
        seed = torch.randint(0, 2**62, (1,))
        if self.world_size > 1:
            from torchdr_amd.parallel import broadcast_

            seed = broadcast_(seed.to(self.device_), src=0).cpu()
        self._neg_seed = int(seed.item())
        self._exclusion = None
        if self.discard_NNs:
            nn_rows = self._nn_for_exclusion()
            if nn_rows is None:
                raise ValueError("[TorchDR] ERROR : discard_NNs=True needs the neighbour table of a sparse affinity "
                                 "(sparsity=True).")
            self_idx = self.chunk_indices_.unsqueeze(1)
            excl = torch.cat([self_idx, nn_rows.to(self_idx.dtype)], dim=1)
            self._exclusion = excl.sort(dim=1).values
        n_possible = self.n_samples_in_ - (1 if self._exclusion is None else self._exclusion.shape[1])
        if self.n_negatives > n_possible and self.verbose:
            raise ValueError(
                f"[TorchDR] ERROR : requested {self.n_negatives} negatives but only {n_possible} available."
            )

    def _nn_for_exclusion(self):
        return self.NN_indices_

    def on_training_step_start(self):
        super().on_training_step_start()
        if self._exclusion is not None:
            w = self._exclusion.shape[1]
            negatives = torch.randint(1, self.n_samples_in_ - w, (self.chunk_size_, self.n_negatives),
                                      device=self.device_)
            shifts = torch.searchsorted(self._exclusion, negatives, right=True)
            self.neg_indices_ = negatives + shifts

    def _neg_ptr_tensor(self):
        """int64 (chunk, n_negatives) table to inject, or None for in-kernel sampling."""
        t = self.neg_indices_
        if t is None:
            return None
        return t.to(device=self.device_, dtype=torch.int64).contiguous()
