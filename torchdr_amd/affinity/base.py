"""Affinity plugin protocol -- mirror of ``torchdr/affinity/base.py`` (reference lines 30, 192, 272, 489).

``Affinity`` / ``LogAffinity`` / ``SparseAffinity`` / ``SparseLogAffinity`` keep the reference's
call contracts: ``aff(X)``, ``aff(X, log=False)``, ``aff(X, return_indices=True) -> (values, indices)``.
An ``AffinityMatcher`` sets ``_pre_processed`` / ``compile`` on the object and reads ``backend``,
``chunk_start_`` and ``chunk_size_`` (affinity_matcher.py:179-181, neighbor_embedding/base.py:393-395).
"""

from typing import Any, Union

import numpy as np
import torch

from torchdr_amd.utils.misc import as_float32
import torch.distributed as dist
import torch.nn as nn

from torchdr_amd.distance import pairwise_distances
from torchdr_amd.distance.base import _pairwise
from torchdr_amd.distributed import DistributedContext
from torchdr_amd.utils import bool_arg, compute_device, set_logger, to_torch


class Affinity(nn.Module):
    def __init__(self, metric: str = "sqeuclidean", zero_diag: bool = True, device: str = "auto", backend=None,
                 verbose: bool = False, random_state: float = None, compile: bool = False,
                 _pre_processed: bool = False):
        super().__init__()
        self.log = {}
        self.metric = metric
        self.zero_diag = bool_arg(zero_diag)
        self.device = device if device is not None else "auto"
        self.backend = backend
        self.verbose = bool_arg(verbose)
        self.random_state = random_state
        self.compile = compile  # accepted for API parity; the HIP kernels need no tracing compiler
        self._pre_processed = _pre_processed
        self.logger = set_logger(self.__class__.__name__, self.verbose)

    def __call__(self, X: Union[torch.Tensor, np.ndarray], **kwargs):
        X = self._prepare(X)
        return self._compute_affinity(X, **kwargs)

    def _prepare(self, X):
        if not self._pre_processed:
            X = to_torch(X)
        # float64 inputs keep their dtype where the float64 kernels cover the affinity (kNN-sparse entropic / UMAP
        # affinities, csrc/tdr_f64.hip), as the reference computes in its input's dtype; everything else runs in float32
        # (row-sharded too: the float64 search takes the rank's chunk of queries, the transposed edges travel in float64)
        if X.dtype == torch.float64 and getattr(self, "_float64_kernels", False):
            return X.to(compute_device(X, self.device))
        return as_float32(X).to(compute_device(X, self.device))

    def _compute_affinity(self, X: torch.Tensor):
        raise NotImplementedError("[TorchDR] ERROR : `_compute_affinity` method is not implemented.")

    def _distance_matrix(self, X: torch.Tensor, k: int = None, return_indices: bool = False):
        return pairwise_distances(X=X, metric=self.metric, backend=self.backend, exclude_diag=self.zero_diag, k=k,
                                  return_indices=return_indices, device=self.device)

    def _get_compute_device(self, X):
        """Reference affinity/base.py:139-160 (tensor or DataLoader input), on this build's HIP device."""
        from torchdr_amd.utils import dataloader_metadata, is_dataloader

        if is_dataloader(X):
            if self.device == "auto":
                return compute_device(torch.empty(0, device=dataloader_metadata(X)[3]), "auto")
            return compute_device(None, self.device)
        return compute_device(X, self.device)

    def _get_n_samples(self, X):
        return X.shape[0]

    def _get_dtype(self, X):
        return X.dtype

    def _get_compute_device(self, X):
        return compute_device(X, self.device)

    def clear_memory(self):
        """Drop non-persistent buffers (reference affinity/base.py:177-189)."""
        for name in list(getattr(self, "_non_persistent_buffers_set", [])):
            if hasattr(self, name):
                delattr(self, name)
        for name in ("_csr_",):
            if hasattr(self, name):
                delattr(self, name)


class LogAffinity(Affinity):
    def __call__(self, X, log: bool = False, **kwargs: Any):
        X = self._prepare(X)
        log_affinity = self._compute_log_affinity(X, **kwargs)
        return log_affinity if log else log_affinity.exp()

    def _compute_log_affinity(self, X: torch.Tensor, **kwargs):
        raise NotImplementedError("[TorchDR] ERROR : `_compute_log_affinity` method is not implemented.")


class SparseAffinity(Affinity):
    """Rectangular (n, k) affinities + kNN indices; row-sharded across ranks when distributed
    (reference affinity/base.py:272-486)."""

    def __init__(self, metric="sqeuclidean", zero_diag=True, device="auto", backend=None, verbose=False,
                 compile=False, sparsity=True, distributed="auto", random_state=None, _pre_processed=False):
        if distributed == "auto":
            self.distributed = dist.is_available() and dist.is_initialized()
        else:
            self.distributed = bool(distributed)
        if self.distributed:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError(
                    "[TorchDR] distributed=True requires launching with torchrun. "
                    "Example: torchrun --nproc_per_node=4 your_script.py"
                )
            self.dist_ctx = DistributedContext()
            self.rank = self.dist_ctx.rank
            self.world_size = self.dist_ctx.world_size
            self.is_multi_gpu = self.world_size > 1
            if device == "cpu":
                raise ValueError("[TorchDR] Distributed mode requires GPU (device cannot be 'cpu')")
            device = torch.device(f"cuda:{self.dist_ctx.local_rank}")
            sparsity = True  # distributed mode is sparse by construction (affinity/base.py:345-348)
        else:
            self.dist_ctx = None
            self.rank = 0
            self.world_size = 1
            self.is_multi_gpu = False
        super().__init__(metric=metric, zero_diag=zero_diag, device=device, backend=backend, verbose=verbose,
                         random_state=random_state, compile=compile, _pre_processed=_pre_processed)
        self.sparsity = sparsity

    @property
    def sparsity(self):
        return self._sparsity

    @sparsity.setter
    def sparsity(self, value):
        self._sparsity = bool_arg(value)

    def __call__(self, X, return_indices: bool = True, **kwargs):
        X = self._prepare(X)
        return self._compute_sparse_affinity(X, return_indices, **kwargs)

    def _compute_sparse_affinity(self, X: torch.Tensor, return_indices: bool = True, **kwargs):
        raise NotImplementedError("[TorchDR] ERROR : `_compute_sparse_affinity` method is not implemented.")

    def _distance_matrix(self, X: torch.Tensor, k: int = None, return_indices: bool = False, info: dict = None):
        """``info``: record shared with the search (``distance.base._pairwise``): the cluster-sorted row order of a pruned
        self search comes back in it, and a row-sharded caller may ask for its rows in that numbering."""
        result = _pairwise(
            X, None, self.metric, self.backend, self.zero_diag, k, return_indices, self.device,
            self.dist_ctx if self.distributed else None, info,
        )
        if self.distributed and self.dist_ctx is not None:
            c0, c1 = self.dist_ctx.compute_chunk_bounds(self._get_n_samples(X))
            self.chunk_start_, self.chunk_end_, self.chunk_size_ = c0, c1, c1 - c0
        return result


class SparseLogAffinity(SparseAffinity, LogAffinity):
    def __call__(self, X, log: bool = False, return_indices: bool = True, **kwargs):
        X = self._prepare(X)
        if return_indices:
            log_affinity, indices = self._compute_sparse_log_affinity(X, return_indices, **kwargs)
            return (log_affinity if log else log_affinity.exp()), indices
        log_affinity = self._compute_sparse_log_affinity(X, return_indices, **kwargs)
        return log_affinity if log else log_affinity.exp()

    def _compute_sparse_log_affinity(self, X: torch.Tensor, return_indices: bool = False, **kwargs):
        raise NotImplementedError(
            "[TorchDR] ERROR : `_compute_sparse_log_affinity` method is not implemented."
        )
