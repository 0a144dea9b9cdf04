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
    if C.dtype == torch.float64:
        return _entropic_search_f64(C.contiguous(), perplexity, n_total, max_iter, use_bounds)
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


def _entropic_search_f64(C, perplexity, n_total, max_iter, use_bounds):
    """float64 twin of :func:`entropic_search` (``tdr_entropic_search_f64``); the scalar p1 root of the bounds is found in
    float64 on the host (entropic.py:51-115)."""
    import math

    n, k = C.shape
    tN, perp = float(n), float(perplexity)
    p1 = 0.5
    if use_bounds:
        max_val = min(math.sqrt(2.0 * tN), perp)
        f = lambda x: math.log(max_val) - 2.0 * (1.0 - x) * math.log(tN / (2.0 * (1.0 - x)))  # noqa: E731
        b, e = 0.75, 1.0 - 1e-6
        # scalar form of root_search.py:17-77 (bounds given, no bracketing needed beyond the reference's two loops)
        for _ in range(1000):
            if not f(b) > 0:
                break
            e, b = min(e, b), b * 0.5
        for _ in range(1000):
            if not f(e) < 0:
                break
            b, e = max(b, e), e * 2.0
        fb, m = f(b), (b + e) * 0.5
        fm = f(m)
        for _ in range(1000):
            if not abs(fm) >= 1e-6:
                break
            if fm * fb > 0:
                b, fb = m, fm
            else:
                e = m
            m = (b + e) * 0.5
            fm = f(m)
        p1 = m
    eps = torch.empty(n, dtype=torch.float64, device=C.device)
    log_norm = torch.empty(n, dtype=torch.float64, device=C.device)
    log_P = torch.empty_like(C)
    _lib.check(
        _lib.lib().tdr_entropic_search_f64(_lib.ptr(C), n, k, math.log(perp) + 1.0, math.log(float(n_total)), int(max_iter), _TOL,
                                           1 if use_bounds else 0, tN, perp, p1, _lib.ptr(eps), _lib.ptr(log_norm), _lib.ptr(log_P),
                                           _lib.stream_ptr()),
        "tdr_entropic_search_f64",
    )
    return eps, log_norm, log_P


class EntropicAffinity(SparseLogAffinity):
    r"""Entropic affinity of SNE / t-SNE: per-row bandwidth :math:`\varepsilon_i` such that the row
    entropy equals :math:`\log(\mathrm{perplexity}) + 1`; rows sum to :math:`1/n`.
    Constructor arguments as in the reference (``entropic.py:196-228``)."""

    _float64_kernels = True   # float64 inputs are computed in float64 (csrc/tdr_f64.hip)

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
            # dense N x N affinity (entropic.py:269-270): the row search streams each full row
            C_, _ = self._distance_matrix(X, return_indices=True)
            eps, log_norm, log_P = entropic_search(C_, int(perplexity), n_samples_in, self.max_iter,
                                                   use_bounds=not self.is_multi_gpu)
            self.register_buffer("eps_", eps, persistent=False)
            self.register_buffer("log_normalization_", log_norm.unsqueeze(1), persistent=False)
            return (log_P, None) if return_indices else log_P
        if self.verbose:
            self.logger.info(f"Sparsity mode enabled, computing {k} nearest neighbors...")
        k = check_neighbor_param(torch.tensor(k), torch.tensor(n_samples_in))
        info = {}
        C_, indices = self._distance_matrix(X, k=int(k), return_indices=True, info=info)
        # the cluster-sorted row order a pruned search worked in (estimators number the points of their loop in it)
        self._row_order = info.get("cluster_order") if not self.is_multi_gpu else None
        eps, log_norm, log_P = entropic_search(
            C_, int(perplexity), n_samples_in, self.max_iter, use_bounds=not self.is_multi_gpu
        )
        self.register_buffer("eps_", eps, persistent=False)
        self.register_buffer("log_normalization_", log_norm.unsqueeze(1), persistent=False)
        return (log_P, indices) if return_indices else log_P


# ------------------------------------------------------------------------------------------------------
# Symmetric entropic affinity (SEA) and Sinkhorn affinity -- matrix-free (reference :315-577, :580-755)
# ------------------------------------------------------------------------------------------------------
from torchdr_amd.affinity.base import LogAffinity  # noqa: E402
from torchdr_amd.distance.base import PackedPoints, dense_packed  # noqa: E402
from torchdr_amd.utils import check_NaNs  # noqa: E402

_DENSE_LIMIT = 30000  # largest N for which the dense (N, N) API output is materialised


def pair_scan_workspace(n: int, n_state: int, device):
    """(pointer, bytes) of the optional split workspace of a pair scan (``tdr_pair_scan_workspace_bytes``) plus the tensor
    that owns it (stream-ordered: allocated and released on the current stream by the caching allocator)."""
    nbytes = int(_lib.lib().tdr_pair_scan_workspace_bytes(n, n_state))
    if nbytes <= 0:
        return None, 0, None
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return _lib.ptr(buf), nbytes, buf


class DensePoints64:
    """float64 input of the symmetric entropic affinity: the DENSE (n, n) float64 matrix of squared distances (``tdr_knn_f64`` with
    k = 0; + 1e12 on an excluded diagonal), which the float64 reductions of ``csrc/tdr_khorn_f64.hip`` read -- the reference
    materialises the same matrix (entropic.py:452).  Parity runs: n <= ``tdr_pairs_f64_max_rows()``."""

    def __init__(self, X: torch.Tensor, zero_diag: bool):
        from torchdr_amd.distance.base import _pairwise_f64

        C = _pairwise_f64(X, None, "sqeuclidean", bool(zero_diag), None, False, "auto", None)
        if C is NotImplemented:
            raise NotImplementedError("[torchdr_amd] float64 SymmetricEntropicAffinity: no float64 distance kernel for this input.")
        self.C, self.n, self.d, self.zero_diag = C, X.shape[0], X.shape[1], bool(zero_diag)

    @staticmethod
    def eligible(X: torch.Tensor) -> bool:
        return (X.dtype == torch.float64 and X.dim() == 2 and X.shape[1] <= 256
                and X.shape[0] <= int(_lib.lib().tdr_pairs_f64_max_rows()))


def pairs64_workspace(n: int, n_state: int, device):
    nbytes = int(_lib.lib().tdr_pairs_f64_workspace_bytes(n, n_state))
    return torch.empty(max(nbytes, 8), dtype=torch.uint8, device=device), nbytes


def sea_rowstats(packed, mu: torch.Tensor, e: torch.Tensor, zero_diag: bool, energy: bool = False):
    """(P_sum, H) of the implicit matrix exp((mu_i+mu_j-2C_ij)/(e_i+e_j)) -- K7 ``tdr_sea_rowstats_f32``; with
    ``energy`` also sum_j P_ij C_ij (the third term of the dual objective, ``tdr_sea_rowstats3_f32``).  float64 input
    (``DensePoints64``): ``tdr_sea_rowstats_dense_f64``."""
    n = packed.n
    if isinstance(packed, DensePoints64):
        if bool(zero_diag) != packed.zero_diag:
            raise ValueError("[torchdr_amd] the dense float64 distance matrix was built for another zero_diag.")
        side = torch.stack([mu.double(), e.double()], dim=1).contiguous()
        out = torch.empty((n, 3), dtype=torch.float64, device=side.device)
        ws, ws_bytes = pairs64_workspace(n, 3, side.device)
        _lib.check(_lib.lib().tdr_sea_rowstats_dense_f64(_lib.ptr(packed.C), n, packed.C.stride(0), _lib.ptr(side), _lib.ptr(out),
                                                        _lib.ptr(ws), ws_bytes, _lib.stream_ptr()), "tdr_sea_rowstats_dense_f64")
        psum, ent = out[:, 0].contiguous(), -(out[:, 1] - out[:, 0])
        return (psum, ent, out[:, 2].contiguous()) if energy else (psum, ent)
    side = torch.stack([mu, e], dim=1).contiguous()
    psum = torch.empty(n, dtype=torch.float32, device=mu.device)
    ent = torch.empty(n, dtype=torch.float32, device=mu.device)
    ws, ws_bytes, _keep = pair_scan_workspace(n, 4, mu.device)
    if energy:
        en = torch.empty(n, dtype=torch.float32, device=mu.device)
        _lib.check(
            _lib.lib().tdr_sea_rowstats3_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(side), 1 if zero_diag else 0,
                                             1e12, _lib.ptr(psum), _lib.ptr(ent), _lib.ptr(en), ws, ws_bytes, _lib.stream_ptr()),
            "tdr_sea_rowstats3_f32",
        )
        return psum, ent, en
    _lib.check(
        _lib.lib().tdr_sea_rowstats_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(side), 1 if zero_diag else 0,
                                        1e12, _lib.ptr(psum), _lib.ptr(ent), ws, ws_bytes, _lib.stream_ptr()),
        "tdr_sea_rowstats_f32",
    )
    return psum, ent


class SymmetricEntropicAffinity(LogAffinity):
    r"""Symmetric entropic affinity of SNEkhorn (reference ``entropic.py:315-577``): dual ascent on
    :math:`(\varepsilon, \mu)` so that :math:`P_{ij} = \exp((\mu_i + \mu_j - 2C_{ij})/(\varepsilon_i^2 +
    \varepsilon_j^2))` has unit row sums and row entropies :math:`\log\xi + 1`.

    The dual loop is **matrix-free**: every iteration recomputes the pairwise distances tile by tile on
    the MFMA pipe and reduces them to the two row statistics the gradients need (no N x N buffer).
    ``fit_duals`` runs just that (what ``TSNEkhorn`` uses); calling the object returns the dense
    log-affinity like the reference does, for N up to ``_DENSE_LIMIT``.  First-order optimizers (reference :518-571)
    and LBFGS (:473-508) both run on the same row statistics."""

    _float64_kernels = True   # float64 inputs: float64 duals on the dense float64 distance matrix (csrc/tdr_khorn_f64.hip), n <= 16384


    def __init__(self, perplexity: float = 30, lr: float = 1e-1, eps_square: bool = True, tol: float = 1e-3,
                 max_iter: int = 500, check_interval: int = 50, optimizer: str = "Adam",
                 metric: str = "sqeuclidean", zero_diag: bool = True, device: str = "auto", backend=None,
                 verbose: bool = False, compile: bool = False, _pre_processed: bool = False):
        super().__init__(metric=metric, zero_diag=zero_diag, device=device, backend=None, verbose=verbose,
                         compile=compile, _pre_processed=_pre_processed)
        self.perplexity = perplexity
        self.lr = lr
        self.eps_square = eps_square
        self.tol = tol
        self.max_iter = max_iter
        self.check_interval = check_interval
        self.optimizer = optimizer
        self.n_iter_ = torch.tensor(0, dtype=torch.long)

    def fit_duals(self, X: torch.Tensor) -> PackedPoints:
        if self.metric != "sqeuclidean":
            raise NotImplementedError("[torchdr_amd] SymmetricEntropicAffinity supports metric='sqeuclidean'.")
        n = X.shape[0]
        # float64 input: float64 duals on the dense float64 distance matrix (the reference computes in its input's dtype);
        # beyond the dense form's size the float32 matrix-free scan, like every other float64 input without a float64 kernel
        if X.dtype == torch.float64 and not DensePoints64.eligible(X):
            X = X.float()
        dt = X.dtype
        packed = DensePoints64(X, self.zero_diag) if dt == torch.float64 else PackedPoints(X)
        perplexity = check_neighbor_param(self.perplexity, n)
        target = torch.log(torch.tensor(float(perplexity), dtype=dt, device=X.device)) + 1
        eps = torch.ones(n, dtype=dt, device=X.device)
        mu = torch.ones(n, dtype=dt, device=X.device)
        self.register_buffer("eps_", eps, persistent=False)
        self.register_buffer("mu_", mu, persistent=False)
        if self.optimizer == "LBFGS":
            return self._fit_duals_lbfgs(packed, target)
        optimizer = getattr(torch.optim, self.optimizer)([self.eps_, self.mu_], lr=self.lr)
        k = 0
        for k in range(self.max_iter):
            with torch.no_grad():
                optimizer.zero_grad()
                e = self.eps_ ** 2 if self.eps_square else self.eps_
                # duals BEFORE this step define the returned matrix (reference returns the pre-step log_P, :573)
                self._dual_snapshot = (self.mu_.clone(), e.clone())
                P_sum, H = sea_rowstats(packed, self.mu_, e, self.zero_diag)
                grad_eps = H - target
                if self.eps_square:
                    grad_eps = 2 * self.eps_.clone().detach() * grad_eps
                grad_mu = P_sum - 1
                self.eps_.grad = grad_eps
                self.mu_.grad = grad_mu
                optimizer.step()
                if not self.eps_square:
                    self.eps_.clamp_(min=0)
                check_NaNs([self.eps_, self.mu_],
                           msg="[TorchDR] ERROR Affinity: NaN at iter {k}, consider decreasing the learning rate.")
                if self.verbose and (k % self.check_interval == 0):
                    perps = (H - 1).exp()
                    self.logger.info(
                        f"[{k}/{self.max_iter}] Perplexity:{float(perps.mean()): .2e} "
                        f"(std:{float(perps.std()): .2e}), Marginal:{float(P_sum.mean()): .2e} "
                        f"(std:{float(P_sum.std()): .2e})"
                    )
                if torch.norm(grad_eps) < self.tol and torch.norm(grad_mu) < self.tol:
                    if self.verbose:
                        self.logger.info(f"Convergence reached at iter {k}.")
                    break
        self.n_iter_ = k
        return packed

    def _fit_duals_lbfgs(self, packed: PackedPoints, target: torch.Tensor) -> PackedPoints:
        """Reference :473-508: ``torch.optim.LBFGS`` (strong Wolfe line search) on the negative Lagrangian.  The reference
        differentiates the loss by autograd; here the closure evaluates loss and gradient from three matrix-free row
        statistics.  The gradient is exact, not an approximation: log P is the minimiser of the Lagrangian for the given
        duals, so the terms that go through P cancel pair by pair ((i, j) against (j, i)) and what is left is
        d/d mu = P_sum - 1, d/d e = H - target (times 2 eps when eps is squared) -- the same expressions the reference
        uses for its first-order optimizers (:529-534)."""
        optimizer = torch.optim.LBFGS([self.eps_, self.mu_], lr=self.lr, max_iter=self.max_iter, tolerance_grad=self.tol,
                                      line_search_fn="strong_wolfe")
        evals = {"n": 0}

        def closure():
            with torch.no_grad():
                e = self.eps_ ** 2 if self.eps_square else self.eps_
                P_sum, H, energy = sea_rowstats(packed, self.mu_, e, self.zero_diag, energy=True)
                loss = -energy.sum() - torch.inner(e, target - H) + torch.inner(self.mu_, P_sum - 1)
                grad_eps = H - target
                if self.eps_square:
                    grad_eps = 2 * self.eps_ * grad_eps
                self.eps_.grad = grad_eps
                self.mu_.grad = P_sum - 1
                evals["n"] += 1
            return loss

        self.eps_.requires_grad_(True)
        self.mu_.requires_grad_(True)
        try:
            optimizer.step(closure)
        finally:
            self.eps_.requires_grad_(False)
            self.mu_.requires_grad_(False)
            self.eps_.grad = None
            self.mu_.grad = None
        check_NaNs([self.eps_, self.mu_],
                   msg="[TorchDR] ERROR Affinity: NaN in dual variables, consider decreasing the learning rate.")
        e = self.eps_ ** 2 if self.eps_square else self.eps_
        self._dual_snapshot = (self.mu_.clone(), e.clone())   # the reference returns log P of the FINAL duals here (:506-508)
        self.n_iter_ = evals["n"]
        return packed

    def dual_side(self):
        """(mu, e) that define the returned affinity: the duals of the LAST evaluated iterate."""
        return self._dual_snapshot

    def _compute_log_affinity(self, X: torch.Tensor):
        n = X.shape[0]
        if n > _DENSE_LIMIT:
            raise MemoryError(
                f"[torchdr_amd] dense SEA output for N={n} would need {4 * n * n / 2**30:.0f} GiB; "
                "use TSNEkhorn (matrix-free) or SymmetricEntropicAffinity.fit_duals."
            )
        packed = self.fit_duals(X)
        mu, e = self.dual_side()
        C = packed.C if isinstance(packed, DensePoints64) else dense_packed(packed, packed, "sqeuclidean", self.zero_diag)
        log_P = (mu[:, None] + mu[None, :] - 2 * C) / (e[:, None] + e[None, :])
        log_P -= math.log(n)
        return log_P


_EMBED_WIDTHS = (2, 3, 4, 8, 16, 32)    # register instances of the embedding-side kernels (tdr_dense.hip)


def pad_embedding(Z: torch.Tensor):
    """(n, nc) -> contiguous fp32 (n, w), w the next kernel width, zero columns added (they change no distance)."""
    Zc = Z.detach().contiguous().float()
    nc = Zc.shape[1]
    if nc > _EMBED_WIDTHS[-1]:
        raise NotImplementedError(f"[torchdr_amd] embedding widths above {_EMBED_WIDTHS[-1]} are not part of the accelerated path.")
    w = next(x for x in _EMBED_WIDTHS if x >= nc)
    if w == nc:
        return Zc
    Zp = torch.zeros((Zc.shape[0], w), dtype=torch.float32, device=Zc.device)
    Zp[:, :nc] = Zc
    return Zp


def sinkhorn_student_dual(Z: torch.Tensor, init_dual, max_iter: int, tol: float, zero_diag: bool = True, record=None):
    """Symmetric Sinkhorn fixed point on the embedding (student kernel, eps = 1): reference
    ``entropic.py:728-748``.  Returns (dual, n_iter).  ``record``: a list that receives, per update that ran,
    (Ef, 1 / s) with Ef = exp(f_before - max) and s_i = sum_j Ef_j / (1 + d_ij) -- what the adjoint of the update needs."""
    _lib.require_gpu(Z, "Z")
    if Z.dtype == torch.float64:
        return _sinkhorn_student_dual64(Z, init_dual, max_iter, tol, zero_diag, record)
    Zc = pad_embedding(Z)
    n, nc = Zc.shape
    f = torch.zeros(n, dtype=torch.float32, device=Z.device) if init_dual is None else init_dual.clone().float()
    f_new = torch.empty_like(f)
    resid2 = torch.zeros(1, dtype=torch.float32, device=Z.device)
    L = _lib.lib()
    ws_bytes = int(L.tdr_student_workspace_bytes(n))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=Z.device)
    k = 0
    for k in range(max_iter):
        fmax = float(f.max())
        Ef = (f - fmax).exp()
        resid2.zero_()
        _lib.check(
            L.tdr_sinkhorn_pass_f32(_lib.ptr(Zc), nc, _lib.ptr(f), _lib.ptr(Ef), fmax, n, 1 if zero_diag else 0,
                                    1e12, _lib.ptr(f_new), _lib.ptr(resid2), _lib.ptr(ws), ws_bytes, _lib.stream_ptr()),
            "tdr_sinkhorn_pass_f32",
        )
        if record is not None:      # f_new = (f - fmax - log s) / 2  =>  1 / s = exp(2 f_new - f + fmax)
            record.append((Ef, (2.0 * f_new - f + fmax).exp()))
            f, f_new = f_new, torch.empty_like(f)
        else:
            f, f_new = f_new, f
        if float(resid2.sqrt()) < tol:
            break
    return f, k


def student_sum64(Z: torch.Tensor, v: torch.Tensor, zero_diag: bool):
    """out_j = sum_i v_i / (1 + |z_j - z_i|^2) in float64 (``tdr_student_sum_f64``; + 1e12 in the diagonal's denominator when
    zero_diag): the reduction of a Student-kernel Sinkhorn update and of its adjoint.  2, 3 or 4 components."""
    n, nc = Z.shape
    side = torch.cat([Z.detach().double(), v.double()[:, None]], dim=1).contiguous()
    out = torch.empty(n, dtype=torch.float64, device=Z.device)
    ws, ws_bytes = pairs64_workspace(n, 1, Z.device)
    _lib.check(_lib.lib().tdr_student_sum_f64(_lib.ptr(side), nc, n, 1 if zero_diag else 0, 1e12, _lib.ptr(out), _lib.ptr(ws), ws_bytes,
                                              _lib.stream_ptr()), "tdr_student_sum_f64")
    return out


def _sinkhorn_student_dual64(Z, init_dual, max_iter, tol, zero_diag, record):
    """float64 form of :func:`sinkhorn_student_dual` (same update, same stopping test, same record)."""
    n = Z.shape[0]
    f = torch.zeros(n, dtype=torch.float64, device=Z.device) if init_dual is None else init_dual.clone().double()
    k = 0
    for k in range(max_iter):
        fmax = float(f.max())
        Ef = (f - fmax).exp()
        s = student_sum64(Z, Ef, zero_diag)
        red = -(fmax + s.log())
        f_new = 0.5 * (f + red)
        if record is not None:
            record.append((Ef, 1.0 / s))
        f = f_new
        if float(torch.norm(f - red)) < tol:
            break
    return f, k


def sinkhorn_student_adjoint(Z: torch.Tensor, record, g_final: torch.Tensor, zero_diag: bool = True, n_terms: int = 5):
    """Reverse sweep through the recorded updates f^k = (f^{k-1} - LSE_j(log K_ij + f^{k-1}_j)) / 2 (reference
    ``entropic.py:733-736`` under ``with_grad=True``): with S^k_ij = Ef^k_j w_ij / s^k_i the softmax update k reduces,
    the adjoints are g^{k-1} = (g^k - S^k^T g^k) / 2 -- one Student-kernel mat-vec each (``tdr_student_matvec_f32``) -- and
    the gradient w.r.t. log K_ij is -(1/2) sum_k g^k_i S^k_ij.  Returns (A, B), both (n, n_terms): A[:, k] = g^k / s^k,
    B[:, k] = Ef^k (zero columns for updates that did not run), so that sum_k g^k_i S^k_ij = w_ij sum_k A_ik B_jk."""
    if Z.dtype == torch.float64:     # the same sweep on the float64 sum (tdr_student_sum_f64)
        n = Z.shape[0]
        A = torch.zeros((n, n_terms), dtype=torch.float64, device=Z.device)
        B = torch.zeros((n, n_terms), dtype=torch.float64, device=Z.device)
        g = g_final.double().contiguous()
        for col, (Ef, inv_s) in enumerate(reversed(record)):
            a = g * inv_s
            A[:, col], B[:, col] = a, Ef
            g = 0.5 * (g - Ef * student_sum64(Z, a, zero_diag))
        return A, B
    Zc = pad_embedding(Z)
    n, nc = Zc.shape
    A = torch.zeros((n, n_terms), dtype=torch.float32, device=Z.device)
    B = torch.zeros((n, n_terms), dtype=torch.float32, device=Z.device)
    g = g_final.float().contiguous()
    t = torch.empty_like(g)
    L = _lib.lib()
    ws_bytes = int(L.tdr_student_workspace_bytes(n))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=Z.device)
    for col, (Ef, inv_s) in enumerate(reversed(record)):
        a = (g * inv_s).contiguous()
        A[:, col], B[:, col] = a, Ef
        _lib.check(L.tdr_student_matvec_f32(_lib.ptr(Zc), nc, _lib.ptr(a), n, 1 if zero_diag else 0, 1e12, _lib.ptr(t),
                                            _lib.ptr(ws), ws_bytes, _lib.stream_ptr()), "tdr_student_matvec_f32")
        g = 0.5 * (g - Ef * t)
    return A, B


def sinkhorn_input_dual(X: torch.Tensor, eps: float, init_dual, max_iter: int, tol: float, zero_diag: bool, student: bool):
    """Symmetric Sinkhorn fixed point on the INPUT points (reference ``entropic.py:693-755``), matrix-free: every
    iteration is one pass of the MFMA pair scan (``tdr_sinkhorn_lse_f32``: distances tile by tile, streaming row
    log-sum-exp), nothing of size N x N exists.  Returns (packed points, dual, n_iter)."""
    _lib.require_gpu(X, "X")
    packed = PackedPoints(X.detach().float())
    n = X.shape[0]
    f = torch.zeros(n, dtype=torch.float32, device=X.device) if init_dual is None else init_dual.clone().float()
    lse = torch.empty_like(f)
    L = _lib.lib()
    k = 0
    ws, ws_bytes, _keep = pair_scan_workspace(n, 2, X.device)
    for k in range(max_iter):
        _lib.check(L.tdr_sinkhorn_lse_f32(_lib.ptr(packed.data), n, packed.d, _lib.ptr(f), 1.0 / float(eps), 1 if student else 0,
                                          1 if zero_diag else 0, 1e12, _lib.ptr(lse), ws, ws_bytes, _lib.stream_ptr()),
                   "tdr_sinkhorn_lse_f32")
        red = -lse
        f = 0.5 * (f + red)
        check_NaNs(f, msg=f"ERROR Affinity: NaN at iter {k}.")
        if float(torch.norm(f - red)) < tol:   # tested AFTER the averaging update (:735-740)
            break
    return packed, f, k


class SinkhornAffinity(LogAffinity):
    r"""Doubly stochastic affinity by symmetric log-domain Sinkhorn (reference ``entropic.py:580-755``):
    :math:`P = \exp(f_i + f_j - C_{ij}/\varepsilon)/N` (``base_kernel="gaussian"``, the class default) or with
    :math:`\log(1 + C)` in place of :math:`C` (``"student"``).  On a 2-D / 3-D input with the student kernel and
    ``eps = 1`` (what TSNEkhorn evaluates on the embedding every step) the update is the LDS-tiled all-pairs kernel;
    any other input runs the matrix-free MFMA pair scan.  ``with_grad=True``: the duals and the returned values are the
    same numbers; there is no autograd graph on the HIP path, so an input that requires grad is refused here --
    ``TSNEkhorn(unrolling=True)``, the reference's one user of the flag, differentiates the updates in closed form
    (``sinkhorn_student_adjoint``)."""

    def __init__(self, eps: float = 1.0, tol: float = 1e-5, max_iter: int = 1000, base_kernel: str = "gaussian",
                 metric: str = "sqeuclidean", zero_diag: bool = True, device: str = "auto", backend=None,
                 verbose: bool = False, with_grad: bool = False, compile: bool = False,
                 _pre_processed: bool = False):
        super().__init__(metric=metric, zero_diag=zero_diag, device=device, backend=None, verbose=verbose,
                         compile=compile, _pre_processed=_pre_processed)
        self.eps = eps
        self.tol = tol
        self.max_iter = max_iter
        self.base_kernel = base_kernel
        self.with_grad = with_grad

    def fit_dual(self, X: torch.Tensor, init_dual=None):
        if self.with_grad and X.requires_grad:
            raise NotImplementedError("[torchdr_amd] SinkhornAffinity(with_grad=True) on an input that requires grad: the HIP "
                                      "path has no autograd graph (TSNEkhorn(unrolling=True) uses the closed-form adjoint).")
        if self.base_kernel not in ("gaussian", "student"):
            raise ValueError(f"[TorchDR] ERROR : base_kernel {self.base_kernel} not supported in SinkhornAffinity.")
        if self.metric != "sqeuclidean":
            raise NotImplementedError("[torchdr_amd] SinkhornAffinity supports metric='sqeuclidean'.")
        if self.base_kernel == "student" and self.eps == 1.0 and X.shape[1] <= 3:
            dual, k = sinkhorn_student_dual(X, init_dual, self.max_iter, self.tol, self.zero_diag)
        else:
            _, dual, k = sinkhorn_input_dual(X, self.eps, init_dual, self.max_iter, self.tol, self.zero_diag,
                                             self.base_kernel == "student")
        self.register_buffer("dual_", dual, persistent=False)
        self.n_iter_ = k
        return dual

    def _compute_log_affinity(self, X: torch.Tensor, init_dual=None):
        n = X.shape[0]
        if n > _DENSE_LIMIT:
            raise MemoryError(f"[torchdr_amd] dense Sinkhorn output for N={n} is not materialised; use fit_dual.")
        dual = self.fit_dual(X, init_dual)
        packed = PackedPoints(X.detach().float())
        C = dense_packed(packed, packed, "sqeuclidean", self.zero_diag)
        if self.base_kernel == "student":
            C = (1 + C).log()
        log_K = -C / self.eps
        return dual[:, None] + dual[None, :] + log_K - math.log(n)
