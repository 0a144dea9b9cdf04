"""Small numeric helpers of ``torchdr/utils/utils.py`` and ``utils/root_search.py`` kept for API parity
(``kmin`` :173, ``kmax`` :219, ``entropy`` :147, ``sum_red`` :303, ``logsumexp_red`` :357,
``cross_entropy_loss`` :100, ``binary_search`` / ``init_bounds`` root_search.py:17,147).

They are host-level glue over tensors the caller already holds (any device); the fused HIP kernels
(``tdr_knn_*``, ``tdr_*_search_f32``) are what the estimators use -- none of these sits on the hot path."""

from typing import Callable, Tuple

import torch


def kmin(A: torch.Tensor, k: int = 1, dim: int = 0):
    """k smallest entries along ``dim`` and their int32 indices; ``(A, None)`` when k >= size (utils.py:203-216)."""
    if not isinstance(dim, int):
        raise ValueError("[TorchDR] ERROR : the input dim to kmin should be an integer.")
    if k >= A.shape[dim]:
        return A, None
    values, indices = A.topk(k=k, dim=dim, largest=False)
    return values, indices.int()


def kmax(A: torch.Tensor, k: int = 1, dim: int = 0):
    if not isinstance(dim, int):
        raise ValueError("[TorchDR] ERROR : the input dim to kmax should be an integer.")
    if k >= A.shape[dim]:
        return A, torch.arange(A.shape[dim]).int()
    values, indices = A.topk(k=k, dim=dim, largest=True)
    return values, indices.int()


def sum_red(P: torch.Tensor, dim):
    """Sum keeping the reduced axes (utils.py:303-354)."""
    if dim is None:
        return P
    if isinstance(dim, int):
        return P.sum(dim, keepdim=True)
    if tuple(dim) == (0, 1):
        return P.sum()
    raise ValueError(f"[TorchDR] ERROR : invalid dim {dim!r} for sum_red.")


def logsumexp_red(log_P: torch.Tensor, dim):
    if dim is None:
        return log_P
    if isinstance(dim, int):
        return log_P.logsumexp(dim, keepdim=True)
    if tuple(dim) == (0, 1):
        return log_P.logsumexp((0, 1))
    raise ValueError(f"[TorchDR] ERROR : invalid dim {dim!r} for logsumexp_red.")


def entropy(P: torch.Tensor, log: bool = True, dim: int = 1):
    """H = -sum P (log P - 1) on probabilities or log-probabilities (utils.py:147-170)."""
    if log:
        return -(P.exp() * (P - 1)).sum(dim).squeeze()
    return -(P * (P.log() - 1)).sum(dim).squeeze()


def cross_entropy_loss(P: torch.Tensor, Q: torch.Tensor, log: bool = False):
    return -sum_red(P * Q, dim=(0, 1)) if log else -sum_red(P * Q.log(), dim=(0, 1))


_TOL = 1e-6


def init_bounds(f: Callable, n: int, begin=1.0, end=1.0, max_iter: int = 100, dtype=torch.float32,
                device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
    """Bracket the roots of an increasing batched function: halve b while f(b) > 0, double e while f(e) < 0."""
    def vec(v):
        if isinstance(v, torch.Tensor):
            v = v.to(dtype=dtype, device=device)
            if v.shape != (n,):
                raise ValueError(f"bound tensor must have shape ({n},), got {v.shape}")
            return v.clone()
        return torch.full((n,), 1.0 if v is None else float(v), dtype=dtype, device=device)

    b, e = vec(begin), vec(end)
    for _ in range(max_iter):
        m = f(b) > 0
        if not m.any():
            break
        e = torch.where(m, torch.minimum(e, b), e)
        b = torch.where(m, b * 0.5, b)
    for _ in range(max_iter):
        m = f(e) < 0
        if not m.any():
            break
        b = torch.where(m, torch.maximum(b, e), b)
        e = torch.where(m, e * 2.0, e)
    return b, e


def binary_search(f: Callable, n: int, begin=1.0, end=1.0, max_iter: int = 100, dtype=torch.float32,
                  device="cpu") -> torch.Tensor:
    """Batched bisection for an arbitrary Python callable (root_search.py:17-77).  The affinities do NOT use
    this: their searches run inside ``tdr_umap_search_f32`` / ``tdr_entropic_search_f32``."""
    tol = torch.tensor(_TOL, dtype=dtype, device=device)
    b, e = init_bounds(f, n, begin, end, max_iter=max_iter, dtype=dtype, device=device)
    f_b = f(b)
    m = (b + e) * 0.5
    f_m = f(m)
    for _ in range(max_iter):
        active = f_m.abs() >= tol
        if not active.any():
            break
        same = f_m * f_b > 0
        go_up = active & same
        b = torch.where(go_up, m, b)
        f_b = torch.where(go_up, f_m, f_b)
        e = torch.where(active & ~same, m, e)
        m = (b + e) * 0.5
        f_m = f(m)
    return m


def false_position(f: Callable, n: int, begin=1.0, end=1.0, max_iter: int = 100, dtype=torch.float32,
                   device="cpu") -> torch.Tensor:
    """Batched regula falsi for an increasing batched function (root_search.py:81-143): the secant through the bracket
    ends replaces the midpoint of `binary_search`; the end whose value has the sign of f(m) moves to m.  Host helper for
    arbitrary callables -- the affinities run their searches inside the HIP kernels."""
    tol = torch.tensor(_TOL, dtype=dtype, device=device)
    b, e = init_bounds(f, n, begin, end, max_iter=max_iter, dtype=dtype, device=device)
    f_b, f_e = f(b), f(e)

    def secant():
        return b - (b - e) / (f_b - f_e) * f_b

    m = secant()
    f_m = f(m)
    for _ in range(max_iter):
        active = f_m.abs() >= tol
        if not active.any():
            break
        same = f_m * f_b > 0
        lower, upper = active & same, active & ~same
        b, f_b = torch.where(lower, m, b), torch.where(lower, f_m, f_b)
        e, f_e = torch.where(upper, m, e), torch.where(upper, f_m, f_e)
        m = secant()
        f_m = f(m)
    return m


def square_loss(P, Q):
    """Sum of squared differences (reference ``utils/utils.py:127-147``)."""
    return ((P - Q) ** 2).sum()


def sum_matrix_vector(M, v, transpose=False):
    """``M + v[:, None]`` (``transpose=False``) or ``M + v[None, :]`` (reference ``utils/utils.py:444-470``)."""
    return M + (v.unsqueeze(-2) if transpose else v.unsqueeze(-1))


def matrix_transpose(arg):
    """Swap the last two axes (reference ``utils/utils.py:526-551``)."""
    if not isinstance(arg, torch.Tensor):
        raise ValueError(f"[TorchDR] ERROR : Unsupported input type for matrix_transpose: {type(arg)}.")
    return arg.transpose(-1, -2)
