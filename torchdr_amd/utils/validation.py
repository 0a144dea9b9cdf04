"""Input checks of the plugin surface.  Only the exception types and message strings are the reference's
(``utils/validation.py:19-28, 223-342``, they are part of the drop-in contract); the checks themselves are organised as a
small rule table, and the finiteness scan of a device-resident block runs in ``tdr_nonfinite_count_f32`` (one fused pass,
one host read) instead of ``torch.isfinite(X).all()``."""

import torch

from torchdr_amd import _lib

_MSG_NAN = "Tensor contains NaN values."
_MSG_INF = "[TorchDR] ERROR : input contains infinite values."


def count_nonfinite(t: torch.Tensor) -> int:
    """Number of inf / nan entries.  float32 matrices on the GPU go through the HIP scan; anything else (host data,
    other dtypes, exotic strides) is counted where it lives."""
    if t.numel() == 0:
        return 0
    if t.is_cuda and t.dtype == torch.float32 and t.dim() in (1, 2) and (t.dim() == 1 or t.stride(1) == 1) and \
            (t.dim() == 1 or t.stride(0) >= t.shape[1]):
        m = t if t.dim() == 2 else t.reshape(1, -1)
        cnt = torch.zeros(1, dtype=torch.int64, device=t.device)
        _lib.check(_lib.lib().tdr_nonfinite_count_f32(_lib.ptr(m), m.shape[0], m.shape[1], m.stride(0), _lib.ptr(cnt),
                                                      _lib.stream_ptr()), "tdr_nonfinite_count_f32")
        return int(cnt.item())
    return int((~torch.isfinite(t)).sum().item())


def check_NaNs(input, msg=None):
    """Raise ``ValueError(msg)`` if a tensor (or any tensor of a list) holds a NaN."""
    tensors = input if isinstance(input, list) else [input]
    for t in tensors:
        if isinstance(t, list):
            check_NaNs(t, msg)
            continue
        if not isinstance(t, torch.Tensor):
            raise TypeError("Input must be a tensor or a list of tensors.")
        if bool(torch.isnan(t).any()):
            raise ValueError(msg or _MSG_NAN)


def check_nonnegativity(P):
    if bool((P < 0).any()):
        raise ValueError("[TorchDR] ERROR : input contains negative values.")


def check_neighbor_param(n_neighbors, n_samples):
    """Neighbour-count style parameters (perplexity, n_neighbors).  Tensor arguments follow the reference's tensor
    branch -- integer truncation, then a clamp into [2, n - 2]; plain numbers are range-checked and returned as they
    are (``validation.py:223-258``)."""
    tensor_form = isinstance(n_neighbors, torch.Tensor) or isinstance(n_samples, torch.Tensor)
    if tensor_form:
        return max(2, min(int(n_neighbors), int(n_samples) - 2))
    if n_samples <= 1:
        raise ValueError(f"[TorchDR] ERROR : Input has less than one sample : n_samples = {n_samples}.")
    if not (1 < n_neighbors < n_samples - 1):
        raise ValueError(
            f"[TorchDR] ERROR : Number of requested neighbors must be greater than "
            f"1 and smaller than the number of samples - 1 (here {n_samples - 1}). "
            f"Got {n_neighbors}."
        )
    return n_neighbors


def _shape_rules(ensure_min_samples, ensure_min_features, max_components):
    """(predicate on (n_samples, n_features), message builder) pairs, evaluated in order."""
    return (
        (lambda n, f: n < ensure_min_samples,
         lambda n, f: f"Found tensor with {n} samples, but a minimum of {ensure_min_samples} is required."),
        (lambda n, f: f < ensure_min_features,
         lambda n, f: f"Found tensor with {f} features, but a minimum of {ensure_min_features} is required."),
        (lambda n, f: max_components is not None and max_components > f,
         lambda n, f: (f"n_components={max_components} is invalid for n_features={f}. "
                       f"The number of components cannot exceed the number of features.")),
    )


def validate_tensor(tensor, accept_sparse=False, ensure_min_samples=1, ensure_min_features=1, ensure_2d=True,
                    max_components=None):
    """Validate an already converted tensor and return it (1-D inputs become a column when ``ensure_2d``)."""
    from .dataloader import is_dataloader

    if is_dataloader(tensor):
        return tensor  # batches are validated while they stream in
    if not isinstance(tensor, torch.Tensor):
        raise ValueError("validate_tensor expects a torch.Tensor, got {}".format(type(tensor)))
    if tensor.is_complex():
        raise ValueError("[TorchDR] ERROR : complex tensors are not supported.")
    if tensor.is_sparse:
        if not accept_sparse:
            raise ValueError("Sparse tensors are not accepted.")
    elif count_nonfinite(tensor):
        raise ValueError(_MSG_INF)
    if ensure_2d:
        if tensor.ndim == 0:
            raise ValueError("Expected 2D tensor, got scalar tensor instead.")
        if tensor.ndim == 1:
            tensor = tensor.reshape(-1, 1)
        elif tensor.ndim != 2:
            raise ValueError(f"Expected 2D tensor, got {tensor.ndim}D tensor instead.")
    n_samples, n_features = tensor.shape
    for violated, message in _shape_rules(ensure_min_samples, ensure_min_features, max_components):
        if violated(n_samples, n_features):
            raise ValueError(message(n_samples, n_features))
    return tensor
