"""K-ary neighbourhood preservation (reference ``eval/neighborhood_preservation.py:15-200``)."""

from typing import Optional, Union

import numpy as np
import torch
import torch.distributed as dist

from torchdr_amd import _lib
from torchdr_amd.distance import pairwise_distances
from torchdr_amd.distributed import DistributedContext
from torchdr_amd.utils.wrappers import to_torch


def _resolve(X, device, distributed):
    """Shared device / distributed handling of the two metrics (reference :111-146)."""
    if distributed == "auto":
        distributed = dist.is_initialized()
    else:
        distributed = bool(distributed)
    if distributed:
        if not dist.is_initialized():
            raise RuntimeError(
                "[TorchDR] distributed=True requires launching with torchrun. "
                "Example: torchrun --nproc_per_node=4 your_script.py"
            )
        ctx = DistributedContext()
        if device == "cpu":
            raise ValueError("[TorchDR] Distributed mode requires GPU (device cannot be 'cpu')")
        device = torch.device(f"cuda:{ctx.local_rank}")
    else:
        ctx = None
        if device is None:
            device = X.device if X.is_cuda else torch.device("cuda", torch.cuda.current_device())
        else:
            device = torch.device(device)
    return device, ctx


def knn_overlap(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Per row, the fraction of the indices of ``a`` (n, K) that also occur in ``b`` (n, K) -- what the
    reference forms as an (n, K, K) broadcast compare (:175-181), here one wavefront per row."""
    _lib.require_gpu(a, "neighbors_X")
    a = a.to(torch.int32).contiguous()
    b = b.to(torch.int32).contiguous()
    n, K = a.shape
    out = torch.empty(n, dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().tdr_knn_overlap_i32(_lib.ptr(a), _lib.ptr(b), n, K, _lib.ptr(out), _lib.stream_ptr()),
               "tdr_knn_overlap_i32")
    return out


def neighborhood_preservation(
    X: Union[torch.Tensor, np.ndarray],
    Z: Union[torch.Tensor, np.ndarray],
    K: int,
    metric: str = "euclidean",
    backend=None,
    device: Optional[str] = None,
    distributed: Union[bool, str] = "auto",
    return_per_sample: bool = False,
):
    """Mean (or per-sample) overlap |N_K^X(i) ∩ N_K^Z(i)| / K of the K nearest neighbours in the input space
    and in the embedding.  Same signature, errors and return types as the reference; both kNN searches run on
    the exact HIP kernel, in distributed mode on this rank's row chunk."""
    input_is_numpy = not isinstance(X, torch.Tensor) or not isinstance(Z, torch.Tensor)
    X = to_torch(X)
    Z = to_torch(Z)
    if X.shape[0] != Z.shape[0]:
        raise ValueError(f"X and Z must have same number of samples, got {X.shape[0]} and {Z.shape[0]}")
    n_samples = X.shape[0]
    if K >= n_samples:
        raise ValueError(f"K ({K}) must be less than number of samples ({n_samples})")
    device, ctx = _resolve(X, device, distributed)
    X = X.to(device=device, dtype=torch.float32)
    Z = Z.to(device=device, dtype=torch.float32)
    _, nx = pairwise_distances(X, metric=metric, backend=backend, k=K, exclude_diag=True, return_indices=True,
                               device=device, distributed_ctx=ctx)
    _, nz = pairwise_distances(Z, metric=metric, backend=backend, k=K, exclude_diag=True, return_indices=True,
                               device=device, distributed_ctx=ctx)
    overlaps = knn_overlap(nx, nz)
    if return_per_sample:
        return overlaps.detach().cpu().numpy() if input_is_numpy else overlaps
    result = overlaps.mean()
    return result.detach().cpu().numpy().item() if input_is_numpy else result
