"""TEST INFRASTRUCTURE ONLY -- plain-torch (CPU) restatement of the reference's hot-path algorithms.

Never imported by the product (``torchdr_amd/``).  Each function cites the reference
file:line (under /root/reference/torchdr) it follows.  Pinned by the golden vectors in
``tests/golden/`` that were produced by importing the real reference in the build container
(``tests/golden/make_golden.py``); see ``tests/test_oracle_golden.py``.
"""

import math

import torch

# --------------------------------------------------------------------------------------------
# kNN -- distance/torch.py:82-122, utils/utils.py:203-216, distance/base.py:183-206 (chunked form)
# --------------------------------------------------------------------------------------------


def knn_chunked(X, k, metric="sqeuclidean", exclude_self=True, Y=None, chunk=4096, rows=None):
    """Row-chunked form of the reference's dense formulation (same arithmetic per element:
    ``(xn + yn) - 2 * (X @ Y.T)`` with MKL sgemm, then topk).  ``rows``: optional (start, stop)
    to compute only a slice of query rows (CPU-baseline sampling)."""
    Yd = X if Y is None else Y
    xn = (X**2).sum(-1)
    yn = xn if Y is None else (Yd**2).sum(-1)
    r0, r1 = (0, X.shape[0]) if rows is None else rows
    outC, outI = [], []
    for s in range(r0, r1, chunk):
        e = min(s + chunk, r1)
        dot = None if metric == "manhattan" else X[s:e] @ Yd.T
        if metric == "manhattan":  # distance/torch.py:96-98 (an (chunk, m, d) intermediate: keep ``chunk`` small)
            C = (X[s:e].unsqueeze(-2) - Yd.unsqueeze(-3)).abs().sum(dim=-1)
        elif metric == "angular":
            C = -dot
        else:
            C = xn[s:e, None] + yn[None, :] - 2 * dot
            if metric == "euclidean":
                C = C.clamp(min=0).sqrt()
            elif metric == "sqhyperbolic":  # distance/torch.py:101-107
                denom = (1 - xn[s:e])[:, None] * (1 - yn)[None, :]
                C = torch.arccosh(1 + 2 * (torch.relu(C) / denom) + 1e-8) ** 2
        if exclude_self and Y is None:
            idx = torch.arange(s, e)
            C[idx - s, idx] = C[idx - s, idx] + 1e12
        v, i = C.topk(k, dim=1, largest=False)
        outC.append(v)
        outI.append(i.int())
    return torch.cat(outC), torch.cat(outI)


def canonical_rows(C, I):
    """Order each row by (distance, index) -- the parity protocol's canonical form."""
    o = torch.argsort(I.long(), dim=1, stable=True)
    C, I = torch.gather(C, 1, o), torch.gather(I, 1, o)
    o = torch.argsort(C, dim=1, stable=True)
    return torch.gather(C, 1, o), torch.gather(I, 1, o)


# --------------------------------------------------------------------------------------------
# Root search -- utils/root_search.py:17-77 (binary_search), :147-198 (init_bounds)
# --------------------------------------------------------------------------------------------

TOL = 1e-6


def init_bounds(f, n, begin=None, end=None, max_iter=100, dtype=torch.float32):
    b = torch.full((n,), 1.0, dtype=dtype) if begin is None else begin.clone().to(dtype)
    e = torch.full((n,), 1.0, dtype=dtype) if end is None else end.clone().to(dtype)
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


def binary_search(f, n, begin=None, end=None, max_iter=100, dtype=torch.float32):
    tol = torch.tensor(TOL, dtype=dtype)
    b, e = init_bounds(f, n, begin, end, max_iter, dtype)
    f_b = f(b)
    m = (b + e) * 0.5
    f_m = f(m)
    for _ in range(max_iter):
        active = f_m.abs() >= tol
        if not active.any():
            break
        same = f_m * f_b > 0
        m1 = active & same
        b = torch.where(m1, m, b)
        f_b = torch.where(m1, f_m, f_b)
        m2 = active & ~same
        e = torch.where(m2, m, e)
        m = (b + e) * 0.5
        f_m = f(m)
    return m


# --------------------------------------------------------------------------------------------
# UMAP affinity -- affinity/knn_normalized.py:417-496 (sparse path, before symmetrisation)
# --------------------------------------------------------------------------------------------


def umap_affinity(C, n_neighbors, max_iter=100):
    """C: (n,k) kNN distances.  Returns rho (n,), eps (n,), P (n,k)."""
    rho = C.min(dim=1).values
    target = torch.log2(torch.tensor(float(n_neighbors), dtype=C.dtype))

    def gap(eps):
        lp = -(C - rho[:, None]) / eps[:, None]
        return lp.logsumexp(1).exp() - target

    eps = binary_search(gap, C.shape[0], max_iter=max_iter, dtype=C.dtype)
    P = (-(C - rho[:, None]) / eps[:, None]).exp()
    return rho, eps, P


# --------------------------------------------------------------------------------------------
# Entropic affinity -- affinity/entropic.py:51-115 (bounds), :230-312 (search + normalisation)
# --------------------------------------------------------------------------------------------


def entropy_log(lp):
    # utils/utils.py:166-168  H = -sum exp(lp) * (lp - 1)
    return -(lp.exp() * (lp - 1)).sum(1)


def entropic_bounds(C, perplexity):
    """Vladymyrov & Carreira-Perpinan bounds as the reference evaluates them (entropic.py:51-115):
    note tN is the number of ROWS of C (n_samples) even when C is the (n,k) kNN block."""
    dtype = C.dtype
    tN = torch.tensor(float(C.shape[0]), dtype=dtype)
    perp = torch.tensor(float(perplexity), dtype=dtype)
    max_val = torch.minimum(torch.sqrt(2.0 * tN), perp)

    def find_p1(x):
        return torch.log(max_val) - 2.0 * (1.0 - x) * torch.log(tN / (2.0 * (1.0 - x)))

    p1 = binary_search(
        find_p1, 1, begin=torch.tensor([0.75], dtype=dtype), end=torch.tensor([1.0 - 1e-6], dtype=dtype),
        max_iter=1000, dtype=dtype,
    ).squeeze()
    dN = C.max(dim=1).values
    d12 = C.topk(2, dim=1, largest=False).values
    d1, d2 = d12[:, 0], d12[:, 1]
    Delta_N = dN - d1
    Delta_2 = d2 - d1
    log_ratio = torch.log(tN / perp)
    beta_L = torch.max((tN * log_ratio) / ((tN - 1) * Delta_N), torch.sqrt(log_ratio / (dN.pow(2) - d1.pow(2))))
    beta_U = (torch.log((tN - 1) * p1 / (1.0 - p1))) / Delta_2
    return 1 / beta_U, 1 / beta_L


def entropic_affinity(C, perplexity, n_total, max_iter=100, use_bounds=True):
    """C: (n,k).  Returns eps (n,), log_norm (n,), log_P (n,k) = -C/eps - LSE - log(n_total)."""
    perp = int(perplexity)
    target = math.log(perp) + 1.0
    target_t = torch.log(torch.tensor(float(perp), dtype=C.dtype)) + 1

    def gap(eps):
        lp = -C / eps[:, None]
        lp = lp - lp.logsumexp(1, keepdim=True)
        return entropy_log(lp) - target_t

    if use_bounds:
        begin, end = entropic_bounds(C, perp)
        begin = begin + 1e-6
    else:
        begin = end = None
    eps = binary_search(gap, C.shape[0], begin=begin, end=end, max_iter=max_iter, dtype=C.dtype)
    lp = -C / eps[:, None]
    log_norm = lp.logsumexp(1, keepdim=True)
    lp = lp - log_norm
    lp = lp - torch.log(torch.tensor(float(n_total), dtype=C.dtype))
    del target
    return eps, log_norm.squeeze(1), lp


# --------------------------------------------------------------------------------------------
# Symmetrisation -- utils/sparse.py:7-206 (flatten_sparse, merge_symmetry, pack_to_rowwise)
# --------------------------------------------------------------------------------------------


def symmetrize_sparse(values, indices, mode="sum_minus_prod"):
    """Q = P + P^T - P o P^T on the union pattern; duplicates summed; rows padded with
    (0, -1), columns ascending.  Returns (values (n,deg), indices int64 (n,deg))."""
    n, k = values.shape
    i = torch.arange(n).repeat_interleave(k)
    j = indices.reshape(-1).long()
    v = values.reshape(-1)
    keys = torch.cat([i * n + j, j * n + i])
    vals = torch.cat([v, v])
    is_p = torch.arange(keys.numel()) < v.numel()
    uniq, inv = torch.unique(keys, sorted=True, return_inverse=True)
    vP = torch.zeros(uniq.numel(), dtype=v.dtype).scatter_add_(0, inv, vals * is_p.to(v.dtype))
    vPT = torch.zeros(uniq.numel(), dtype=v.dtype).scatter_add_(0, inv, vals * (~is_p).to(v.dtype))
    out = vP + vPT if mode == "sum" else vP + vPT - vP * vPT
    io, jo = uniq // n, uniq % n
    counts = torch.bincount(io, minlength=n)
    deg = int(counts.max().item()) if io.numel() else 0
    V = torch.zeros((n, deg), dtype=v.dtype)
    J = torch.full((n, deg), -1, dtype=torch.long)
    offs = torch.zeros(n + 1, dtype=torch.long)
    offs[1:] = counts.cumsum(0)
    slot = torch.arange(io.numel()) - offs[io]
    V[io, slot] = out
    J[io, slot] = jo
    return V, J


# --------------------------------------------------------------------------------------------
# UMAP loop -- neighbor_embedding/umap.py:215-292, neighbor_embedding/base.py:617-649
# --------------------------------------------------------------------------------------------


def umap_prepare(A, max_iter):
    """umap.py:215-234: epochs_per_sample = A_max / (A + 1e-3), inf where A <= A_max / max_iter."""
    A_max = A.max()
    small = A <= A_max / max_iter
    eps_per = (A + 1e-3).reciprocal() * A_max
    eps_per = eps_per.masked_fill(small, float("inf"))
    return eps_per, eps_per.clone()


def umap_gradients(Z, NN, eps_per, next_, neg, n_iter, a, b, neg_rate=5, eps=1e-3, rows=None):
    """One evaluation of umap.py:236-292.  Z (N,c); NN (n,K) int64 (-1 pads wrap to row N-1, as
    PyTorch indexing does in the reference); neg (n, n_neg) int64.  Mutates ``next_`` in place.
    Returns (grad_attr, grad_rep, active_mask)."""
    rows = torch.arange(NN.shape[0]) if rows is None else rows
    Zi = Z[rows]
    diff = Zi[:, None, :] - Z[NN]
    D = (diff**2).sum(-1)
    pos = D > 0
    D_ = 1 + a * D**b
    coef = D.pow(b - 1) * (2 * a * b) / D_
    coef = coef.masked_fill(~pos, 0)
    act = next_ <= n_iter + 1
    next_[act] += eps_per[act]
    coef = coef.masked_fill(~act, 0)
    g_attr = torch.einsum("ijk,ij->ik", diff, coef).clamp(-4, 4)

    diffn = Zi[:, None, :] - Z[neg]
    Dn = (diffn**2).sum(-1)
    Dn_ = 1 + a * Dn**b
    cn = ((Dn + eps) * Dn_).reciprocal() * (-2 * b)
    cnt = (act.sum(1) * neg_rate).long()
    col = torch.arange(neg.shape[1])
    cn = cn.masked_fill(col[None, :] >= cnt[:, None], 0)
    g_rep = torch.einsum("ijk,ij->ik", diffn, cn).clamp(-4, 4)
    return g_attr, g_rep, act


def sample_negatives(n_total, rows, n_neg, generator=None):
    """neighbor_embedding/base.py:628-636: randint(0, N-1) then +1 where >= own index."""
    r = torch.randint(0, n_total - 1, (rows.numel(), n_neg), generator=generator)
    return r + (r >= rows[:, None]).long()


# --------------------------------------------------------------------------------------------
# LargeVis / TSNE -- neighbor_embedding/largevis.py:181-201, tsne.py:162-180 (closed forms of
# the autograd gradients, SURVEY.md appendix A.4, verified against autograd in the tests)
# --------------------------------------------------------------------------------------------


def ne_attraction_grad(Z, NN, P, kind, rows=None):
    """d/dZ of  -sum_ij P_ij log Q_ij ;  largevis: Q = 1/(2+d), tsne / infotsne: log Q = -log(1+d),
    sne: log Q = -d (sne.py:160-168).  Both endpoints of every edge receive gradient."""
    rows = torch.arange(NN.shape[0]) if rows is None else rows
    NN = NN.long()
    diff = Z[rows][:, None, :] - Z[NN]
    D = (diff**2).sum(-1)
    w = 2 * P if kind == "sne" else 2 * P / ((2 + D) if kind == "largevis" else (1 + D))
    g = torch.zeros_like(Z)
    contrib = w[:, :, None] * diff
    g.index_add_(0, rows, contrib.sum(1))
    g.index_add_(0, NN.reshape(-1), -contrib.reshape(-1, Z.shape[1]))
    return g


def largevis_repulsion_grad(Z, neg, n_total, rows=None):
    """d/dZ of  -(1/N) sum log(1 - Q), Q = 1/(2+d)  ->  w = -(2/N) / ((1+d)(2+d))."""
    rows = torch.arange(neg.shape[0]) if rows is None else rows
    diff = Z[rows][:, None, :] - Z[neg]
    D = (diff**2).sum(-1)
    w = -(2.0 / n_total) / ((1 + D) * (2 + D))
    g = torch.zeros_like(Z)
    contrib = w[:, :, None] * diff
    g.index_add_(0, rows, contrib.sum(1))
    g.index_add_(0, neg.reshape(-1), -contrib.reshape(-1, Z.shape[1]))
    return g


def tsne_repulsion_grad(Z):
    """d/dZ of log sum_ij (1+d_ij)^-1 (diagonal included, tsne.py:172-180):
    g_i = -(4/S) sum_j (z_i - z_j) / (1+d_ij)^2."""
    D = torch.cdist(Z.double(), Z.double()) ** 2
    W = 1.0 / (1.0 + D)
    S = W.sum()
    W2 = W * W
    g = -(4.0 / S) * (W2.sum(1, keepdim=True) * Z.double() - W2 @ Z.double())
    return g.to(Z.dtype), S.to(Z.dtype)


def sne_repulsion_grad(Z):
    """d/dZ of (1/N) sum_i logsumexp_j(-d_ij) (dense, diagonal included; sne.py:170-179):
    g_i = -(2/N) sum_j e_ij (1/R_i + 1/R_j) (z_i - z_j), e = exp(-d), R_i = sum_j e_ij."""
    Zd = Z.double()
    E = torch.exp(-(torch.cdist(Zd, Zd) ** 2))
    invR = 1.0 / E.sum(1)
    W = E * (invR[:, None] + invR[None, :])
    g = -(2.0 / Z.shape[0]) * (W.sum(1, keepdim=True) * Zd - W @ Zd)
    return g.to(Z.dtype)


def infotsne_repulsion_grad(Z, neg, n_total, rows=None):
    """d/dZ of (1/N) sum_i log sum_{n in Neg(i)} q_in, q = 1/(1+d) (infotsne.py:188-197):
    edge weight w_in = -(2/N) q_in^2 / sum_n q_in on (z_i - z_n); both endpoints move."""
    rows = torch.arange(neg.shape[0]) if rows is None else rows
    diff = Z[rows][:, None, :] - Z[neg]
    Q = 1.0 / (1.0 + (diff**2).sum(-1))
    w = -(2.0 / n_total) * Q * Q / Q.sum(1, keepdim=True)
    g = torch.zeros_like(Z)
    contrib = w[:, :, None] * diff
    g.index_add_(0, rows, contrib.sum(1))
    g.index_add_(0, neg.reshape(-1), -contrib.reshape(-1, Z.shape[1]))
    return g


def pacmap_affinity(X, n_neighbors):
    """affinity/knn_normalized.py:574-611 (PACMAPAffinity): the n_neighbors + 50 nearest by squared distance
    (self excluded), rho_i = mean Euclidean distance to the 4th-6th, rescale by rho_i * rho_j, keep the n_neighbors
    smallest.  Returns (indices (n, n_neighbors) int64, rho)."""
    n = X.shape[0]
    k = min(n_neighbors + 50, n)
    C, I = knn_chunked(X, k, "sqeuclidean", True)
    rho = torch.sqrt(C[:, :6])[:, 3:6].mean(dim=1)
    Cs = C / (rho[:, None] * rho[I.long()])
    local = torch.topk(Cs, n_neighbors, dim=1, largest=False).indices
    return torch.gather(I.long(), 1, local), rho


def pacmap_grad(Z, near, mid, far, w_nb, w_mn, w_fp):
    """d/dZ of PaCMAP's pair losses (neighbor_embedding/pacmap.py:213-265), q = 1 + d:
    w_nb sum q/(10+q) over near pairs, w_mn sum q/(1e4+q) over mid-near pairs, w_fp sum 1/(1+q) over further pairs;
    d/dd = 10 w_nb/(11+d)^2, 1e4 w_mn/(10001+d)^2, -w_fp/(2+d)^2.  Both endpoints of every pair move."""
    g = torch.zeros_like(Z)
    rows = torch.arange(Z.shape[0])
    for idx, num, off, w in ((near, 10.0, 11.0, w_nb), (mid, 1.0e4, 10001.0, w_mn), (far, -1.0, 2.0, w_fp)):
        if idx is None or w == 0:
            continue
        idx = idx.long()
        diff = Z[rows][:, None, :] - Z[idx]
        D = (diff**2).sum(-1)
        coef = 2.0 * w * num / (off + D) ** 2
        contrib = coef[:, :, None] * diff
        g.index_add_(0, rows, contrib.sum(1))
        g.index_add_(0, idx.reshape(-1), -contrib.reshape(-1, Z.shape[1]))
    return g


def sgd_momentum_step(Z, grad, buf, lr, momentum):
    """torch.optim.SGD (no dampening / nesterov / weight decay): buf = momentum*buf + grad
    (first step: buf = grad); Z -= lr * buf."""
    if momentum == 0:
        return Z - lr * grad, buf
    buf = grad.clone() if buf is None else buf * momentum + grad
    return Z - lr * buf, buf


# --------------------------------------------------------------------------------------------
# TSNEkhorn -- affinity/entropic.py:437-577 (SEA, first-order optimizer branch), :693-755 (Sinkhorn,
# student kernel), neighbor_embedding/tsnekhorn.py:210-230 (loss; gradient in closed form)
# --------------------------------------------------------------------------------------------


def sea_affinity(C, perplexity, lr=1e-1, max_iter=100, tol=1e-3, eps_square=True, optimizer="Adam"):
    """C: dense (n, n) squared distances (diagonal as the caller wants it).  Returns
    (eps, mu, log_P - log n, n_iter); log_P is the matrix evaluated BEFORE the last dual step (:573)."""
    n = C.shape[0]
    target = torch.log(torch.tensor(float(perplexity), dtype=C.dtype)) + 1
    eps = torch.ones(n, dtype=C.dtype)
    mu = torch.ones(n, dtype=C.dtype)
    opt = getattr(torch.optim, optimizer)([eps, mu], lr=lr)
    log_P = None
    k = 0
    for k in range(max_iter):
        e = eps**2 if eps_square else eps
        log_P = (mu[:, None] + mu[None, :] - 2 * C) / (e[:, None] + e[None, :])
        H = entropy_log(log_P)
        P_sum = log_P.logsumexp(1).exp()
        g_eps = H - target
        if eps_square:
            g_eps = 2 * eps.clone() * g_eps
        g_mu = P_sum - 1
        eps.grad, mu.grad = g_eps, g_mu
        opt.step()
        if not eps_square:
            eps.clamp_(min=0)
        if torch.norm(g_eps) < tol and torch.norm(g_mu) < tol:
            break
    return eps, mu, log_P - math.log(n), k


def sea_dual_objective(C, eps, mu, perplexity, eps_square=True):
    """The closure of SymmetricEntropicAffinity(optimizer="LBFGS") (affinity/entropic.py:483-491): negative Lagrangian
    -sum(P C) - <e, target - H> + <mu, rowsum - 1> with log P = (mu_i + mu_j - 2 C_ij) / (e_i + e_j), e = eps^2 or eps.
    Returns (loss, H, rowsum); its gradients are H - target (times 2 eps when squared) and rowsum - 1."""
    e = eps**2 if eps_square else eps
    log_P = (mu[:, None] + mu[None, :] - 2 * C) / (e[:, None] + e[None, :])
    P = log_P.exp()
    H = -(P * (log_P - 1)).sum(1)
    rowsum = P.sum(1)
    target = math.log(perplexity) + 1
    loss = -(P * C).sum() - torch.inner(e, target - H) + torch.inner(mu, rowsum - 1)
    return loss, H, rowsum


def sinkhorn_student(Z, init_dual=None, max_iter=5, tol=1e-5, zero_diag=True):
    n = Z.shape[0]
    D = torch.cdist(Z, Z) ** 2 if False else ((Z[:, None, :] - Z[None, :, :]) ** 2).sum(-1)
    if zero_diag:
        D = D + torch.diag(torch.full((n,), 1e12, dtype=Z.dtype))
    log_K = -(1 + D).log()
    dual = torch.zeros(n, dtype=Z.dtype) if init_dual is None else init_dual.clone()
    k = 0
    for k in range(max_iter):
        red = -(log_K + dual[:, None]).logsumexp(0)
        dual = 0.5 * (dual + red)
        if torch.norm(dual - red) < tol:
            break
    return dual, log_K, k


def tsnekhorn_grad(Z, log_P, dual, log_K):
    """Closed-form gradient of CE(P, log Q) + sum(Q) with the dual detached:
    4 sum_j (P_ij - Q_ij)/(1 + d_ij) (z_i - z_j);  log Q = dual_i + dual_j + log K - log n."""
    n = Z.shape[0]
    log_Q = dual[:, None] + dual[None, :] + log_K - math.log(n)
    W = log_K.exp()  # 1/(1+d)
    M = (log_P.exp() - log_Q.exp()) * W
    return 4 * (M.sum(1, keepdim=True) * Z - M @ Z)


def tsnekhorn_unrolled_grad(Z, log_P, init_dual=None, max_iter=5, tol=1e-5, zero_diag=True):
    """TSNEkhorn(unrolling=True) (neighbor_embedding/tsnekhorn.py:134, 183, 224-227): loss = CE(P, log Q) alone, log Q =
    f_i + f_j + log K - log n with f the result of the <= max_iter Sinkhorn updates of affinity/entropic.py:733-743 started
    from the DETACHED warm start and run under autograd (with_grad=True).  Restated with torch autograd on dense matrices,
    exactly as the reference evaluates it.  Returns (gradient w.r.t. Z, dual, n_iter)."""
    n = Z.shape[0]
    Zg = Z.detach().clone().requires_grad_(True)
    D = ((Zg[:, None, :] - Zg[None, :, :]) ** 2).sum(-1)
    if zero_diag:
        D = D + torch.diag(torch.full((n,), 1e12, dtype=Z.dtype))
    log_K = -(1 + D).log()
    dual = torch.zeros(n, dtype=Z.dtype) if init_dual is None else init_dual.detach().clone()
    k = 0
    for k in range(max_iter):
        red = -(log_K + dual[:, None]).logsumexp(0)
        dual = 0.5 * (dual + red)
        if torch.norm(dual - red) < tol:
            break
    log_Q = dual[:, None] + dual[None, :] + log_K - math.log(n)
    loss = -(log_P.exp() * log_Q).sum()
    (g,) = torch.autograd.grad(loss, Zg)
    return g, dual.detach(), k


def tsnekhorn_unrolled_grad_closed(Z, log_P, init_dual=None, max_iter=5, tol=1e-5, zero_diag=True):
    """The same gradient without autograd -- the formulation the HIP path evaluates (csrc/tdr_dense.hip, KhornForceUnrolled):
    adjoints g^K = -(rowsum P + colsum P), g^{k-1} = (g^k - S^k^T g^k) / 2 with S^k the softmax update k reduces, and
    grad_i = sum_j [4 P_ij + w_ij sum_k (a^k_i b^k_j + a^k_j b^k_i)] w_ij (z_i - z_j), a^k = g^k / s^k, b^k = exp(f^{k-1} - max)."""
    n = Z.shape[0]
    D = ((Z[:, None, :] - Z[None, :, :]) ** 2).sum(-1)
    if zero_diag:
        D = D + torch.diag(torch.full((n,), 1e12, dtype=Z.dtype))
    W = 1 / (1 + D)
    f = torch.zeros(n, dtype=Z.dtype) if init_dual is None else init_dual.clone()
    rec = []
    for _ in range(max_iter):
        fmax = f.max()
        Ef = (f - fmax).exp()
        s = W @ Ef
        red = -(fmax + s.log())
        f = 0.5 * (f + red)
        rec.append((Ef, s))
        if torch.norm(f - red) < tol:
            break
    P = log_P.exp()
    g = -(P.sum(1) + P.sum(0))
    bil = torch.zeros_like(W)
    for Ef, s in reversed(rec):
        a = g / s
        bil = bil + a[:, None] * Ef[None, :] + Ef[:, None] * a[None, :]
        g = 0.5 * (g - Ef * (W @ a))
    coef = (4 * P + W * bil) * W
    return coef.sum(1, keepdim=True) * Z - coef @ Z, f


# --------------------------------------------------------------------------------------------
# COSNE -- neighbor_embedding/cosne.py:162-193 (loss), utils/manifold.py:207-330 (Poincare ball),
# utils/radam.py:96-167 (Riemannian Adam), affinity_matcher.py:552-565 (hyperbolic init).  float64 throughout.
# --------------------------------------------------------------------------------------------


def cosne_loss(Z, P, NN, X_norm, gamma, lam, exag=1.0, rep=1.0, chunk_start=0):
    """The reference's scalar loss (autograd-able): exag * CE(P, log Q on the kNN graph, gathered distances by direct
    difference, distance/base.py:392-398) + rep * (log sum_ij Q_ij over the dense norm-expansion matrix,
    distance/torch.py:82-107, + lam * mean_i (||x_i||^2 - d_H(z_i, 0)^2)^2)."""
    n_rows = P.shape[0]
    zn = (Z**2).sum(-1)
    Zq = Z[chunk_start:chunk_start + n_rows]
    Yk = Z[NN.long()]
    d = torch.relu(((Zq[:, None, :] - Yk) ** 2).sum(-1))
    den = (1 - zn[chunk_start:chunk_start + n_rows, None]) * (1 - (Yk**2).sum(-1))
    d = torch.arccosh(1 + 2 * (d / den) + 1e-8) ** 2
    attr = -(P * (gamma / (d + gamma**2)).log()).sum()
    C = torch.relu(zn[:, None] + zn[None, :] - 2 * Z @ Z.T)
    D = torch.arccosh(1 + 2 * (C / ((1 - zn)[:, None] * (1 - zn)[None, :])) + 1e-8) ** 2
    rep_loss = (gamma / (D + gamma**2)).log().logsumexp((0, 1))
    h = torch.arccosh(1 + 2 * (zn / (1 - zn)) + 1e-8) ** 2
    return exag * attr + rep * (rep_loss + lam * ((X_norm - h) ** 2).mean())


def _dd2_dzi(zi, zo):
    """d/dz_i of d_H(z_i, z_o)^2 = arccosh(1 + 2 s / (a_i a_o) + 1e-8)^2; returns (d2, grad) for broadcastable inputs."""
    s = ((zi - zo) ** 2).sum(-1)
    ai, ao = 1 - (zi**2).sum(-1), 1 - (zo**2).sum(-1)
    w = 1 + 2 * s / (ai * ao) + 1e-8
    u = torch.arccosh(w)
    du = 2 * u / torch.sqrt(w * w - 1)                              # d(u^2)/dw
    dw = 4 * (zi - zo) / (ai * ao)[..., None] + (4 * s / (ai * ai * ao))[..., None] * zi
    return u * u, du[..., None] * dw


def cosne_grad(Z, P, NN, X_norm, gamma, lam, exag=1.0, rep=1.0):
    """Closed-form Euclidean gradient of :func:`cosne_loss` (single process): what the HIP kernels evaluate."""
    n, c = Z.shape
    G = torch.zeros_like(Z)
    # attraction: both ends of every directed edge (i -> j)
    Zi = Z[:, None, :].expand(-1, NN.shape[1], -1)
    Zj = Z[NN.long()]
    d2, gi = _dd2_dzi(Zi, Zj)
    _, gj = _dd2_dzi(Zj, Zi)
    wgt = (P.to(Z.dtype) / (d2 + gamma**2))[..., None]
    G += exag * (wgt * gi).sum(1)
    G.index_add_(0, NN.reshape(-1).long(), exag * (wgt * gj).reshape(-1, c))
    # repulsion: log sum_ij Q_ij, Q = gamma / (d2 + gamma^2), diagonal contributes to the sum only
    D2, gI = _dd2_dzi(Z[:, None, :], Z[None, :, :])
    Q = gamma / (D2 + gamma**2)
    S = Q.sum()
    dQ = -(gamma / (D2 + gamma**2) ** 2)[..., None] * gI
    dQ[torch.arange(n), torch.arange(n)] = 0
    G += rep * (2.0 / S) * dQ.sum(1)
    # norm preservation
    y = (Z**2).sum(-1)
    w = 1 + 2 * y / (1 - y) + 1e-8
    u = torch.arccosh(w)
    dh = (2 * u / torch.sqrt(w * w - 1)) * (2 / (1 - y) ** 2)
    G += rep * lam * (2.0 / n) * ((u * u - X_norm) * dh)[:, None] * (2 * Z)
    return G


def _lambda_x(x):
    return 2 / (1.0 - (x**2).sum(-1, keepdim=True)).clamp_min(1e-15)


def _mobius_add(x, y):
    x2, y2, xy = (x**2).sum(-1, keepdim=True), (y**2).sum(-1, keepdim=True), (x * y).sum(-1, keepdim=True)
    num = (1 + 2 * xy + y2) * x + (1 - x2) * y
    return num / (1 + 2 * xy + x2 * y2).clamp_min(1e-15)


def radam_poincare_step(Z, egrad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8):
    """One RiemannianAdam step on the unit Poincare ball (c = 1, no weight decay / amsgrad), utils/radam.py:139-167.
    ``step`` is the group's counter, which the reference increments TWICE per step.  Returns
    (Z_new, exp_avg_new, exp_avg_sq_new, step_new, rgrad)."""
    lam = _lambda_x(Z)
    g = egrad / lam**2                                                    # egrad2rgrad
    exp_avg = exp_avg * betas[0] + (1 - betas[0]) * g
    exp_avg_sq = exp_avg_sq * betas[1] + (1 - betas[1]) * (lam**2 * (g * g).sum(-1, keepdim=True))   # inner, keepdim
    denom = exp_avg_sq.sqrt() + eps
    step += 1
    step_size = lr * (1 - betas[1] ** step) ** 0.5 / (1 - betas[0] ** step)
    u = -step_size * (exp_avg / denom)
    un = u.norm(dim=-1, keepdim=True).clamp_min(1e-15)
    second = (0.5 * lam * un).clamp(-15, 15).tanh() * u / un               # expmap
    new = _mobius_add(Z, second)
    nn_ = new.norm(dim=-1, keepdim=True).clamp_min(1e-15)                  # proj (float64: eps 1e-5)
    maxnorm = 1 - 1e-5
    new = torch.where(nn_ > maxnorm, new / nn_ * maxnorm, new)
    # ptransp(point, new_point, exp_avg): gyration(new, -point, exp_avg) * lambda_point / lambda_new
    uu, vv, ww = new, -Z, exp_avg
    u2, v2 = (uu**2).sum(-1, keepdim=True), (vv**2).sum(-1, keepdim=True)
    uv, uw, vw = (uu * vv).sum(-1, keepdim=True), (uu * ww).sum(-1, keepdim=True), (vv * ww).sum(-1, keepdim=True)
    a = -uw * v2 + vw + 2 * uv * vw
    b = -vw * u2 - uw
    d = 1 + 2 * uv + u2 * v2
    gyr = ww + 2 * (a * uu + b * vv) / d.clamp_min(1e-15)
    exp_avg = gyr * lam / _lambda_x(new)
    step += 1
    return new, exp_avg, exp_avg_sq, step, g


def hyperbolic_init(noise, init_scaling=0.5):
    """affinity_matcher.py:552-565: expmap0(init_scaling * randn) on the unit ball (float64)."""
    u = init_scaling * noise
    un = u.norm(dim=-1, keepdim=True).clamp_min(1e-15)
    return un.clamp(-15, 15).tanh() * u / un


# --------------------------------------------------------------------------------------------
# Approximate search -- distance/faiss.py:331-349 (IndexIVFFlat: nlist inverted lists, nprobe probed per query)
# --------------------------------------------------------------------------------------------


def ivf_search(X, row_map, tile_cluster, clus_dist, nprobe, k, metric="sqeuclidean", exclude_self=True, group_tiles=4):
    """Deterministic restatement of an inverted-file search on a GIVEN index (Faiss is absent from the image; the reference
    hands ``FaissConfig(index_type="IVF", nlist, nprobe)`` to ``faiss.IndexIVFFlat``, distance/faiss.py:331-349, whose
    published algorithm is: assign every point to its nearest centroid's list, probe the ``nprobe`` lists whose centroids are
    nearest, return the exact top-k over the union of the probed lists).

    The index is the input (``row_map``: list-sorted, tile-padded row order, -1 = padding; ``tile_cluster``: list of every
    32-row tile; ``clus_dist``: (nlist, nlist) centre distances) -- the restatement does not cluster.  Probe rule, as
    documented for ``tdr_knn_ivf_f32``: queries are taken in blocks of ``group_tiles`` consecutive 32-row tiles of the sorted
    order; a block probes the lists of its own tiles (all of them together are probe 1) and then the ``nprobe - 1`` other lists
    in increasing (distance of the list's centre to the NEAREST of the block's own centres, list id) order.  Result per query:
    exact top-k over the rows of the probed lists with the reference's arithmetic (``oracle.knn``: distance/torch.py:82-122)
    in canonical (distance, index) order, self excluded (distance/torch.py:111-116); a query that finds fewer than k
    candidates is searched exactly against all rows (Faiss would pad with -1; the package never lets -1 reach the affinity
    stages).  Returns (C (n, k) float32, I (n, k) int32, short (n,) bool = rows that took the exact fallback)."""
    import oracle

    X = X.detach().cpu().contiguous().float()
    row_map = row_map.detach().cpu().long()
    tile_cluster = tile_cluster.detach().cpu().long()
    clus_dist = clus_dist.detach().cpu().float()
    n = X.shape[0]
    nlist = clus_dist.shape[0]
    n_tiles = tile_cluster.numel()
    img_cluster = tile_cluster.repeat_interleave(32)[: row_map.numel()]
    valid = row_map >= 0
    cluster_of_row = torch.empty(n, dtype=torch.long)
    cluster_of_row[row_map[valid]] = img_cluster[valid]
    members = [torch.sort((cluster_of_row == c).nonzero().squeeze(1)).values for c in range(nlist)]
    C = torch.empty((n, k), dtype=torch.float32)
    I = torch.empty((n, k), dtype=torch.int32)
    short = torch.zeros(n, dtype=torch.bool)
    bits = clus_dist.clamp(min=0).contiguous().view(torch.int32).long()      # the kernel orders by the float's bit pattern
    for t0 in range(0, n_tiles, group_tiles):
        own = torch.unique(tile_cluster[t0:t0 + group_tiles])
        key = bits[own].min(0).values                                         # (nlist,) distance to the nearest own centre
        order = sorted(range(nlist), key=lambda c: (int(key[c]), c))
        probed, others = [], 0
        for c in order:
            if int(key[c]) == 0:
                probed.append(c)                                              # own lists (and centres that coincide with one)
            elif others < nprobe - 1:
                probed.append(c)
                others += 1
        cand = torch.sort(torch.cat([members[c] for c in probed])).values   # ascending source index: ties break as in the full search
        q = row_map[t0 * 32:(t0 + group_tiles) * 32]
        q = q[q >= 0]
        if q.numel() == 0:
            continue
        Yc = X[cand]
        # one call per block: top-(k + 1) WITHOUT exclusion in canonical order, then the query's own row is taken out of its
        # list (dropping one entry of a canonical top-(k + 1) leaves the canonical top-k of the rest)
        pos = torch.searchsorted(cand, q)
        member = (pos < cand.numel()) & (cand[pos.clamp(max=cand.numel() - 1)] == q)
        drop_self = member & bool(exclude_self)
        avail = cand.numel() - drop_self.long()
        ok = avail >= k
        short[q[~ok]] = True
        if not bool(ok.any()):
            continue
        kk = min(k + 1, cand.numel())
        c1, i1 = oracle.knn(X[q[ok]], kk, metric, False, Y=Yc)
        src = cand[i1.long()]
        qs = q[ok]
        keep = src != qs[:, None] if exclude_self else torch.ones_like(src, dtype=torch.bool)
        for r in range(qs.numel()):
            sel = keep[r].nonzero().squeeze(1)[:k]
            C[qs[r]] = c1[r, sel]
            I[qs[r]] = src[r, sel].int()
    rows = short.nonzero().squeeze(1)
    for qi in rows.tolist():
        c1, i1 = oracle.knn(X[qi:qi + 1], k, metric, exclude_self, Y=X, q_offset=qi)
        C[qi], I[qi] = c1[0], i1[0]
    return C, I, short
