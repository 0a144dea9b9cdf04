"""sklearn-style estimator shell -- mirror of ``torchdr/base.py`` (``DRModule``, reference :19-229)."""

from abc import ABC, abstractmethod
from typing import Any, Optional

import torch

from torchdr_amd.utils.misc import as_float32
import torch.nn as nn
from sklearn.base import BaseEstimator

from torchdr_amd.utils import handle_input_output, seed_everything, set_logger


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

    @handle_input_output()
    def fit(self, X, y: Optional[Any] = None):
        self.fit_transform(X, y=y)
        return self

    @handle_input_output()
    def fit_transform(self, X, y: Optional[Any] = None):
        """Fit and return the embedding.  Duplicate rows are embedded once and re-expanded
        (reference base.py:132-148)."""
        in_dtype = X.dtype
        X = as_float32(X)  # float64 in -> computed in float32 -> float64 out
        if self.process_duplicates:
            X_unique, inverse = torch.unique(X, dim=0, return_inverse=True)
            if X_unique.shape[0] < X.shape[0]:
                self.logger.info(
                    f"Detected {X.shape[0] - X_unique.shape[0]} duplicate samples, performing DR on unique data."
                )
                emb = self._fit_transform(X_unique, y=y)
                self.embedding_ = emb[inverse.to(emb.device)]
            else:
                self.embedding_ = self._fit_transform(X, y=y)
        else:
            self.embedding_ = self._fit_transform(X, y=y)
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

    def clear_memory(self):
        for name in list(getattr(self, "_non_persistent_buffers_set", [])):
            if hasattr(self, name):
                delattr(self, name)
