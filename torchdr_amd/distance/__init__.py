"""``torchdr.distance`` surface.  One backend -- the HIP kernels behind ``pairwise_distances`` -- answers under every name the
reference exports: its per-backend entry points (``distance/torch.py:21``, ``distance/faiss.py:225,477``) keep their argument
order and return convention ((distances, indices) always; indices ``None`` without ``k``) and select the same search."""

from typing import Any, Optional

import torch

from .base import (  # noqa: F401
    LIST_METRICS,
    PackedPoints,
    knn_packed,
    dense_packed,
    pairwise_distances,
    pairwise_distances_indexed,
)
from .faiss import FaissConfig  # noqa: F401

LIST_METRICS_TORCH = list(LIST_METRICS)
LIST_METRICS_FAISS = ["euclidean", "sqeuclidean", "angular"]


def pairwise_distances_torch(X: torch.Tensor, Y: torch.Tensor = None, metric: str = "sqeuclidean", k: int = None,
                             exclude_diag: bool = False, device: str = "auto"):
    return pairwise_distances(X, Y, metric=metric, backend=None, exclude_diag=exclude_diag, k=k, return_indices=True,
                              device=device)


def _faiss_metric(metric):
    if metric not in LIST_METRICS_FAISS:
        raise ValueError(f"[TorchDR] Only {LIST_METRICS_FAISS} metrics are supported for FAISS.")


def pairwise_distances_faiss(X: torch.Tensor, k, Y: torch.Tensor = None, metric: str = "sqeuclidean",
                             exclude_diag: bool = False, config: Optional[FaissConfig] = None, device: str = "auto"):
    """``config`` with ``index_type="IVF"`` / ``"IVFPQ"`` selects the approximate search, anything else the exact one."""
    _faiss_metric(metric)
    return pairwise_distances(X, Y, metric=metric, backend=config if config is not None else "faiss",
                              exclude_diag=exclude_diag, k=int(k), return_indices=True, device=device)


def pairwise_distances_faiss_from_dataloader(dataloader, k: int, metric: str = "sqeuclidean", exclude_diag: bool = False,
                                             config: Optional[FaissConfig] = None, device: str = "auto",
                                             distributed_ctx: Optional[Any] = None):
    _faiss_metric(metric)
    return pairwise_distances(dataloader, metric=metric, backend=config if config is not None else "faiss",
                              exclude_diag=exclude_diag, k=int(k), return_indices=True, device=device,
                              distributed_ctx=distributed_ctx)
