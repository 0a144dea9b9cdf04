"""k-NN label accuracy (reference ``eval/knn_labels.py:17-190``): the share of each point's k nearest neighbours that carry
the point's own label.  The neighbour search is the exact HIP kNN (K1 / K1s); only the argument errors are the
reference's strings."""

from typing import Optional, Union

import numpy as np
import torch

from torchdr_amd.distance import pairwise_distances
from torchdr_amd.eval.neighborhood_preservation import _resolve
from torchdr_amd.utils.wrappers import to_torch


def _label_agreement(labels: torch.Tensor, idx: torch.Tensor, first_row: int) -> torch.Tensor:
    """Per query row r (global index first_row + r): mean over its neighbours of [label(neighbour) == label(query)]."""
    own = labels[first_row:first_row + idx.shape[0]]
    return (labels[idx.long()] == own[:, None]).to(torch.float32).mean(dim=1)


def knn_label_accuracy(X: Union[torch.Tensor, np.ndarray], labels: Union[torch.Tensor, np.ndarray], k: int = 10,
                       metric: str = "euclidean", backend="faiss", exclude_self: bool = True,
                       distributed: Union[bool, str] = "auto", return_per_sample: bool = False,
                       device: Optional[str] = None):
    """Mean (or per-sample) k-NN label accuracy; under a distributed context each rank scores its own row chunk
    (reference :172-178).  numpy in -> python float / numpy array out, tensors in -> tensors out."""
    as_numpy = not (isinstance(X, torch.Tensor) and isinstance(labels, torch.Tensor))
    if k < 1:
        raise ValueError(f"k must be at least 1, got {k}")
    X, labels = to_torch(X), to_torch(labels)
    n = X.shape[0]
    if labels.shape[0] != n:
        raise ValueError(f"X and labels must have same number of samples, got {n} and {labels.shape[0]}")
    if k >= n:
        raise ValueError(f"k ({k}) must be less than number of samples ({n})")
    device, ctx = _resolve(X, device, distributed)
    _, idx = pairwise_distances(X.to(device=device, dtype=torch.float32), metric=metric, backend=backend, k=k,
                                exclude_diag=exclude_self, return_indices=True, device=device, distributed_ctx=ctx)
    first_row = ctx.compute_chunk_bounds(n)[0] if (ctx is not None and ctx.is_initialized) else 0
    scores = _label_agreement(labels.to(device), idx, first_row)
    if not return_per_sample:
        scores = scores.mean()
    if as_numpy:
        scores = scores.detach().cpu().numpy()
        return scores if return_per_sample else scores.item()
    return scores
