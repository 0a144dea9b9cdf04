"""sklearn-style estimator shell -- mirror of ``torchdr/base.py`` (``DRModule``, reference :19-229)."""

from abc import ABC, abstractmethod
from typing import Any, Optional

import torch

from torchdr_amd.utils.misc import as_float32
import torch.nn as nn
from sklearn.base import BaseEstimator

from torchdr_amd.utils import compute_device, handle_input_output, seed_everything, set_logger


def unique_rows(X: torch.Tensor, device="auto"):
    """Duplicate-row removal of ``fit_transform`` (reference base.py:133: ``torch.unique(X, dim=0, return_inverse=True)``).
    Returns ``(X_unique, inverse)`` with ``inverse = None`` when every row is unique (the common case: X is returned as it
    is, on the compute device).  Duplicates are FOUND by 64-bit row hashes in an open-addressing table plus an exact
    row compare (``tdr_dedup_rows_f32``: one pass over X instead of the lexicographic merge sort of the N x D block);
    the unique rows keep their original order (the reference's sorted order only relabels the points: the embedding is
    re-expanded through ``inverse`` either way).  A 64-bit hash collision between different rows (probability ~N^2/2^65)
    falls back to the exact sort."""
    from torchdr_amd import _lib
    from torchdr_amd.utils import compute_device

    if X.dim() == 2 and X.dtype == torch.float64 and X.shape[0] >= 2:
        # float64 rows that are equal are equal after rounding to float32: the hash pass on the rounded block finds every
        # candidate; only when it reports duplicates (or collisions) is the float64 block itself sorted
        X = X.to(compute_device(X, device))
        _, inv32 = unique_rows(X.to(torch.float32), device)
        if inv32 is None:
            return X, None
        X_unique, inverse = torch.unique(X, dim=0, return_inverse=True)
        return (X_unique, inverse) if X_unique.shape[0] < X.shape[0] else (X, None)
    if X.dim() != 2 or X.dtype != torch.float32 or X.shape[0] < 2:
        return X, None
    X = X.to(compute_device(X, device))
    if X.stride(1) != 1:
        X = X.contiguous()
    n, d = X.shape
    L = _lib.lib()
    ws_bytes = int(L.tdr_dedup_workspace_bytes(n))
    ws = torch.empty(ws_bytes // 8 + 1, dtype=torch.int64, device=X.device)
    rep = torch.empty(n, dtype=torch.int32, device=X.device)
    counters = torch.zeros(2, dtype=torch.int32, device=X.device)
    _lib.check(L.tdr_dedup_rows_f32(_lib.ptr(X), n, d, X.stride(0), _lib.ptr(rep), _lib.ptr(counters), _lib.ptr(ws), ws_bytes,
                                    _lib.stream_ptr()), "tdr_dedup_rows_f32")
    n_dup, n_collide = (int(v) for v in counters.tolist())
    if n_collide:
        X_unique, inverse = torch.unique(X, dim=0, return_inverse=True)
        return (X_unique, inverse) if X_unique.shape[0] < n else (X, None)
    if n_dup == 0:
        return X, None
    keep = rep == torch.arange(n, dtype=torch.int32, device=X.device)
    new_index = keep.long().cumsum(0) - 1
    return X[keep], new_index[rep.long()]


class DRModule(BaseEstimator, nn.Module, ABC):
    def __init__(self, n_components: int = 2, device: str = "auto", backend=None, verbose: bool = False,
                 random_state: Optional[float] = None, compile: bool = False, process_duplicates: bool = True,
                 **kwargs):
        super().__init__()
        self.n_components = n_components
        self.device = device if device is not None else "auto"
        self.backend = backend
        self.verbose = verbose
        self.random_state = random_state
        self.compile = compile
        self.process_duplicates = process_duplicates
        self.logger = set_logger(self.__class__.__name__, self.verbose)
        if self.random_state is not None:
            self._actual_seed = seed_everything(self.random_state, fast=True, deterministic=False)
            self.logger.info(f"Random seed set to: {self._actual_seed}.")
        self.embedding_ = None
        self.is_fitted_ = False

    def _float64_ok(self, X) -> bool:
        """Estimator-specific limits of the float64 kernels (`_float64_loop` estimators): False sends a float64 input down
        the float32 path (computed in float32, handed back as float64) instead of failing on an unsupported shape."""
        return True

    @handle_input_output()
    def fit(self, X, y: Optional[Any] = None):
        self.fit_transform(X, y=y)
        return self

    @handle_input_output()
    def fit_transform(self, X, y: Optional[Any] = None):
        """Fit and return the embedding.  Duplicate rows are embedded once and re-expanded
        (reference base.py:132-148)."""
        in_dtype = X.dtype
        # float64 in: estimators with float64 kernels for the whole path (`_float64_loop`: UMAP, LargeVis, TSNE, InfoTSNE, SNE,
        # PaCMAP; D <= 256; single process or row-sharded) compute in float64 like the reference (which computes in its
        # input's dtype); the others compute in float32 and hand float64 back
        if not (X.dtype == torch.float64 and getattr(self, "_float64_loop", False) and self._float64_ok(X)
                and X.dim() == 2 and X.shape[1] <= 256 and getattr(self, "metric", "sqeuclidean") in ("sqeuclidean", "euclidean", "angular")):
            X = as_float32(X)
        if getattr(self, "sharded_input", False) and getattr(self, "world_size", 1) > 1:
            # X is this rank's row shard: one all-gather of the shards, then the replicated-input path
            from torchdr_amd.parallel import gather_row_shards
            from torchdr_amd.utils.phases import phase

            with phase("shard gather (all-gather of X)"):
                X = gather_row_shards(X)
        inverse = None
        if self.process_duplicates:
            from torchdr_amd.utils.phases import phase

            with phase("input checks (dedup)"):
                X, inverse = unique_rows(X, self.device)
            if inverse is not None:
                self.logger.info(
                    f"Detected {inverse.numel() - X.shape[0]} duplicate samples, performing DR on unique data."
                )
        emb = self._fit_transform(X, y=y)
        self.embedding_ = emb if inverse is None else emb[inverse.to(emb.device)]
        if in_dtype == torch.float64:
            self.embedding_ = self.embedding_.to(torch.float64)
        self.is_fitted_ = True
        return self.embedding_

    def transform(self, X=None):
        if not self.is_fitted_:
            raise ValueError(
                "This DRModule instance is not fitted yet. Call 'fit' or 'fit_transform' with some data first."
            )
        if X is not None:
            raise NotImplementedError("Transforming new data is not implemented for this model.")
        return self.embedding_

    @abstractmethod
    def _fit_transform(self, X: torch.Tensor, y: Optional[Any] = None) -> torch.Tensor:
        raise NotImplementedError("[TorchDR] ERROR : _fit_transform method is not implemented.")

    def _get_compute_device(self, X):
        """Reference base.py:215-217, with this build's rule that compute happens on a HIP device (utils.compute_device)."""
        return compute_device(X, self.device)

    def clear_memory(self):
        for name in list(getattr(self, "_non_persistent_buffers_set", [])):
            if hasattr(self, name):
                delattr(self, name)
