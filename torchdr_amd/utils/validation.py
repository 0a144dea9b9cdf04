"""Input checks with the reference's messages (``utils/validation.py:19-28, 223-342``)."""

import torch


def check_NaNs(input, msg=None):
    if isinstance(input, list):
        for t in input:
            check_NaNs(t, msg)
    elif isinstance(input, torch.Tensor):
        if torch.isnan(input).any():
            raise ValueError(msg or "Tensor contains NaN values.")
    else:
        raise TypeError("Input must be a tensor or a list of tensors.")


def check_nonnegativity(P):
    if (P < 0).any():
        raise ValueError("[TorchDR] ERROR : input contains negative values.")


def check_neighbor_param(n_neighbors, n_samples):
    """validation.py:223-258.  Tensor arguments are clamped (and TRUNCATED to integers) into
    [2, n-2]; plain numbers are range-checked and returned unchanged."""
    if isinstance(n_neighbors, torch.Tensor) or isinstance(n_samples, torch.Tensor):
        n = int(n_samples)
        k = int(n_neighbors)  # long-tensor coercion truncates (perplexity 30.7 -> 30)
        return max(2, min(k, n - 2))
    if n_samples <= 1:
        raise ValueError(f"[TorchDR] ERROR : Input has less than one sample : n_samples = {n_samples}.")
    if n_neighbors <= 1 or n_neighbors >= n_samples - 1:
        raise ValueError(
            f"[TorchDR] ERROR : Number of requested neighbors must be greater than "
            f"1 and smaller than the number of samples - 1 (here {n_samples - 1}). "
            f"Got {n_neighbors}."
        )
    return n_neighbors


def validate_tensor(tensor, accept_sparse=False, ensure_min_samples=1, ensure_min_features=1, ensure_2d=True,
                    max_components=None):
    """validation.py:261-342."""
    if not isinstance(tensor, torch.Tensor):
        raise ValueError("validate_tensor expects a torch.Tensor, got {}".format(type(tensor)))
    if torch.is_complex(tensor):
        raise ValueError("[TorchDR] ERROR : complex tensors are not supported.")
    if not tensor.is_sparse and not torch.isfinite(tensor).all():
        raise ValueError("[TorchDR] ERROR : input contains infinite values.")
    if not accept_sparse and tensor.is_sparse:
        raise ValueError("Sparse tensors are not accepted.")
    if ensure_2d:
        if tensor.ndim == 0:
            raise ValueError("Expected 2D tensor, got scalar tensor instead.")
        elif tensor.ndim == 1:
            tensor = tensor.reshape(-1, 1)
        if tensor.ndim != 2:
            raise ValueError(f"Expected 2D tensor, got {tensor.ndim}D tensor instead.")
    n_samples, n_features = tensor.shape
    if n_samples < ensure_min_samples:
        raise ValueError(
            f"Found tensor with {n_samples} samples, but a minimum of {ensure_min_samples} is required."
        )
    if n_features < ensure_min_features:
        raise ValueError(
            f"Found tensor with {n_features} features, but a minimum of {ensure_min_features} is required."
        )
    if max_components is not None and max_components > n_features:
        raise ValueError(
            f"n_components={max_components} is invalid for n_features={n_features}. "
            f"The number of components cannot exceed the number of features."
        )
    return tensor
