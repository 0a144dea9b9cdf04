"""k-NN label accuracy (reference ``eval/knn_labels.py:17-190``)."""

from typing import Optional, Union

import numpy as np
import torch

from torchdr_amd.distance import pairwise_distances
from torchdr_amd.eval.neighborhood_preservation import _resolve
from torchdr_amd.utils.wrappers import to_torch


def knn_label_accuracy(
    X: Union[torch.Tensor, np.ndarray],
    labels: Union[torch.Tensor, np.ndarray],
    k: int = 10,
    metric: str = "euclidean",
    backend="faiss",
    exclude_self: bool = True,
    distributed: Union[bool, str] = "auto",
    return_per_sample: bool = False,
    device: Optional[str] = None,
):
    """Fraction of each point's k nearest neighbours that carry the point's own label (mean or per sample);
    reference semantics incl. the distributed chunk (each rank scores its rows, ``knn_labels.py:172-178``)."""
    if k < 1:
        raise ValueError(f"k must be at least 1, got {k}")
    input_is_numpy = not isinstance(X, torch.Tensor) or not isinstance(labels, torch.Tensor)
    X = to_torch(X)
    labels = to_torch(labels)
    if X.shape[0] != labels.shape[0]:
        raise ValueError(
            f"X and labels must have same number of samples, got {X.shape[0]} and {labels.shape[0]}"
        )
    n_samples = X.shape[0]
    if k >= n_samples:
        raise ValueError(f"k ({k}) must be less than number of samples ({n_samples})")
    device, ctx = _resolve(X, device, distributed)
    X = X.to(device=device, dtype=torch.float32)
    labels = labels.to(device)
    _, idx = pairwise_distances(X, metric=metric, backend=backend, k=k, exclude_diag=exclude_self,
                                return_indices=True, device=device, distributed_ctx=ctx)
    neighbor_labels = labels[idx.long()]
    if ctx is not None and ctx.is_initialized:
        c0, c1 = ctx.compute_chunk_bounds(n_samples)
        query_labels = labels[c0:c1].unsqueeze(1)
    else:
        query_labels = labels.unsqueeze(1)
    accuracies = (neighbor_labels == query_labels).float().mean(dim=1)
    if return_per_sample:
        return accuracies.detach().cpu().numpy() if input_is_numpy else accuracies
    result = accuracies.mean()
    return result.detach().cpu().numpy().item() if input_is_numpy else result
