"""Entropic (perplexity-calibrated) affinities on the GPU -- mirror of ``torchdr/affinity/entropic.py``.

* ``EntropicAffinity``  (reference :118-312)  -- kNN + per-row bandwidth search (K2).
"""

import math
from typing import Union

import torch

from torchdr_amd import _lib
from torchdr_amd.affinity.base import SparseLogAffinity
from torchdr_amd.utils import check_neighbor_param

_TOL = 1e-6  # utils/root_search.py:13


def _scalar_binary_search(f, begin: float, end: float, max_iter: int = 1000) -> torch.Tensor:
    """The 1-D case of utils/root_search.py:17-77,147-198 on fp32 scalars (used for the p1 root of
    the Vladymyrov bounds, entropic.py:78-94).  O(1) host work, same arithmetic as the reference."""
    b = torch.tensor(begin, dtype=torch.float32)
    e = torch.tensor(end, dtype=torch.float32)
    for _ in range(max_iter):
        if not (f(b) > 0):
            break
        e = torch.minimum(e, b)
        b = b * 0.5
    for _ in range(max_iter):
        if not (f(e) < 0):
            break
        b = torch.maximum(b, e)
        e = e * 2.0
    f_b = f(b)
    m = (b + e) * 0.5
    f_m = f(m)
    for _ in range(max_iter):
        if not (f_m.abs() >= _TOL):
            break
        if f_m * f_b > 0:
            b, f_b = m, f_m
        else:
            e = m
        m = (b + e) * 0.5
        f_m = f(m)
    return m


def entropic_bound_scalars(n_rows: int, perplexity: int):
    """Scalar part of ``_bounds_entropic_affinity`` (entropic.py:51-115); note ``tN`` is the number
    of rows of C (the reference passes the (n, k) block and still uses n)."""
    tN = torch.tensor(float(n_rows), dtype=torch.float32)
    perp = torch.tensor(float(perplexity), dtype=torch.float32)
    max_val = torch.minimum(torch.sqrt(2.0 * tN), perp)

    def find_p1(x):
        return torch.log(max_val) - 2.0 * (1.0 - x) * torch.log(tN / (2.0 * (1.0 - x)))

    p1 = _scalar_binary_search(find_p1, 0.75, 1.0 - 1e-6, max_iter=1000)
    log_ratio = torch.log(tN / perp)
    return {
        "tN_logratio": float(tN * log_ratio),
        "tN_m1": float(tN - 1),
        "log_ratio": float(log_ratio),
        "beta_u_num": float(torch.log((tN - 1) * p1 / (1.0 - p1))),
    }


def entropic_search(C: torch.Tensor, perplexity: int, n_total: int, max_iter: int, use_bounds: bool = True):
    """eps_i with H(softmax(-C_i/eps_i)) = log(perp) + 1; returns (eps, log_norm, log_P) with
    log_P = -C/eps - LSE - log(n_total)  (entropic.py:272-310, K2 ``tdr_entropic_search_f32``)."""
    _lib.require_gpu(C, "C")
    C = C.contiguous().float()
    n, k = C.shape
    eps = torch.empty(n, dtype=torch.float32, device=C.device)
    log_norm = torch.empty(n, dtype=torch.float32, device=C.device)
    log_P = torch.empty_like(C)
    target = float(torch.log(torch.tensor(float(perplexity), dtype=torch.float32)) + 1)
    log_n = float(torch.log(torch.tensor(float(n_total), dtype=torch.float32)))
    sc = entropic_bound_scalars(n, perplexity) if use_bounds else dict(tN_logratio=0.0, tN_m1=0.0, log_ratio=0.0,
                                                                       beta_u_num=0.0)
    _lib.check(
        _lib.lib().tdr_entropic_search_f32(
            _lib.ptr(C), n, k, target, log_n, int(max_iter), _TOL, 1 if use_bounds else 0, sc["tN_logratio"],
            sc["tN_m1"], sc["log_ratio"], sc["beta_u_num"], _lib.ptr(eps), _lib.ptr(log_norm), _lib.ptr(log_P),
            _lib.stream_ptr(),
        ),
        "tdr_entropic_search_f32",
    )
    return eps, log_norm, log_P


class EntropicAffinity(SparseLogAffinity):
    r"""Entropic affinity of SNE / t-SNE: per-row bandwidth :math:`\varepsilon_i` such that the row
    entropy equals :math:`\log(\mathrm{perplexity}) + 1`; rows sum to :math:`1/n`.
    Constructor arguments as in the reference (``entropic.py:196-228``)."""

    def __init__(self, perplexity: float = 30, max_iter: int = 1000, sparsity: bool = True,
                 metric: str = "sqeuclidean", zero_diag: bool = True, device: str = "auto",
                 backend: Union[str, None] = None, verbose: bool = False, compile: bool = False,
                 distributed: Union[bool, str] = "auto", _pre_processed: bool = False):
        self.perplexity = perplexity
        self.max_iter = max_iter
        super().__init__(metric=metric, zero_diag=zero_diag, device=device, backend=backend, verbose=verbose,
                         sparsity=sparsity, compile=compile, distributed=distributed,
                         _pre_processed=_pre_processed)

    def _compute_sparse_log_affinity(self, X: torch.Tensor, return_indices: bool = True, **kwargs):
        n_samples_in = self._get_n_samples(X)
        # tensor form of check_neighbor_param: integer truncation + clamp to [2, n-2] (validation.py:229-244)
        perplexity = check_neighbor_param(torch.tensor(self.perplexity), torch.tensor(n_samples_in))
        k = 3 * perplexity
        if not self.sparsity:
            raise NotImplementedError(
                "[torchdr_amd] EntropicAffinity(sparsity=False) (dense N x N affinity) is not part of the "
                "accelerated path; use sparsity=True."
            )
        if self.verbose:
            self.logger.info(f"Sparsity mode enabled, computing {k} nearest neighbors...")
        k = check_neighbor_param(torch.tensor(k), torch.tensor(n_samples_in))
        C_, indices = self._distance_matrix(X, k=int(k), return_indices=True)
        eps, log_norm, log_P = entropic_search(
            C_, int(perplexity), n_samples_in, self.max_iter, use_bounds=not self.is_multi_gpu
        )
        self.register_buffer("eps_", eps, persistent=False)
        self.register_buffer("log_normalization_", log_norm.unsqueeze(1), persistent=False)
        return (log_P, indices) if return_indices else log_P
