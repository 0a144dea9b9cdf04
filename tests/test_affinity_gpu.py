"""K2 / K3 / K4 parity (GPU): root searches and symmetrisation vs golden vectors of the real reference."""

import os

import numpy as np
import pytest
import torch

from tests.conftest import gmm
from tests.test_oracle_golden import load

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # north_star tolerance for affinities (relative, fp32)


def test_umap_sigma_search_vs_reference():
    from torchdr_amd.affinity.knn_normalized import umap_sigma_search

    g = load("affinity")
    for nn in (10, 30):
        rho, eps, P = umap_sigma_search(g[f"umap{nn}_C"].cuda(), nn, 100)
        assert torch.equal(rho.cpu(), g[f"umap{nn}_rho"])
        assert torch.allclose(eps.cpu(), g[f"umap{nn}_eps"], rtol=RTOL, atol=0)
        assert torch.allclose(P.cpu(), g[f"umap{nn}_P"], rtol=RTOL, atol=1e-8)


def test_entropic_search_vs_reference():
    from torchdr_amd.affinity.entropic import entropic_search

    g = load("affinity")
    n = g["X"].shape[0]
    for perp in (5, 30):
        eps, lognorm, logP = entropic_search(g[f"ent{perp}_C"].cuda(), perp, n, 100)
        assert torch.allclose(eps.cpu(), g[f"ent{perp}_eps"], rtol=RTOL, atol=0)
        ref = g[f"ent{perp}_logP"]
        assert torch.allclose(logP.cpu(), ref, rtol=RTOL, atol=RTOL * float(ref.abs().max()))
        assert torch.allclose(logP.exp().cpu(), ref.exp(), rtol=1e-4, atol=1e-9)
        # invariant the reference's own tests pin: row entropy == log(perp) + 1 (test_affinity.py:209-210)
        p = (logP + np.log(n)).exp()
        H = -(p * (p.log() - 1)).sum(1)
        assert torch.allclose(H.cpu(), torch.full((n,), float(np.log(perp) + 1)), atol=1e-3)
    # multi-GPU style search without bounds reaches the same root
    eps_nb, _, _ = entropic_search(g["ent30_C"].cuda(), 30, n, 100, use_bounds=False)
    assert torch.allclose(eps_nb.cpu(), g["ent30_eps"], rtol=1e-4)


def test_large_k_searches():
    """k > 32 / > 64 / > 128 code paths against the plain-torch oracle."""
    from oracle import ref_torch as R
    from torchdr_amd.affinity.entropic import entropic_search
    from torchdr_amd.affinity.knn_normalized import umap_sigma_search

    gen = torch.Generator().manual_seed(3)
    for k in (45, 90, 200):
        C = (torch.rand(300, k, generator=gen) * 5 + 0.1).sort(1).values
        rho, eps, P = umap_sigma_search(C.cuda(), k, 100)
        r2, e2, P2 = R.umap_affinity(C, k, 100)
        assert torch.equal(rho.cpu(), r2) and torch.allclose(eps.cpu(), e2, rtol=RTOL)
        assert torch.allclose(P.cpu(), P2, rtol=RTOL, atol=1e-8)
        perp = k // 3
        e, ln, lp = entropic_search(C.cuda(), perp, 300, 100)
        e2, ln2, lp2 = R.entropic_affinity(C, perp, 300, 100)
        assert torch.allclose(e.cpu(), e2, rtol=RTOL)
        assert torch.allclose(lp.cpu(), lp2, rtol=RTOL, atol=1e-5)


def test_symmetrize_vs_reference():
    from torchdr_amd.utils.sparse import symmetrize_sparse, symmetrize_to_csr

    g = load("symmetrize")
    V, J = symmetrize_sparse(g["vals"].cuda(), g["idx"].cuda())
    assert J.dtype == torch.int64
    assert torch.equal(J.cpu(), g["J"])
    assert torch.allclose(V.cpu(), g["V"], rtol=0, atol=1e-7)
    V, J = symmetrize_sparse(g["vals"].cuda(), g["idx"].cuda(), mode="sum")
    assert torch.equal(J.cpu(), g["J_sum"]) and torch.allclose(V.cpu(), g["V_sum"], rtol=0, atol=1e-7)
    a = load("affinity")
    for nn in (10, 30):
        V, J = symmetrize_sparse(a[f"umap{nn}_P"].cuda(), a[f"umap{nn}_I"].cuda())
        assert torch.equal(J.cpu(), a[f"umap{nn}_Isym"])
        assert torch.equal(V.cpu(), a[f"umap{nn}_Psym"])  # same op order => bit-identical
        csr = symmetrize_to_csr(a[f"umap{nn}_P"].cuda(), a[f"umap{nn}_I"].cuda())
        assert csr.nnz == int((a[f"umap{nn}_Isym"] >= 0).sum())


def test_symmetrize_rows_of_every_length_class_vs_oracle():
    """Hub columns: symmetrised rows of 20 .. ~2500 entries -- the one-entry-per-lane sort, the register-resident forms of 2 / 4 / 8 /
    16 entries per lane (65 .. 1024 entries) and the general path beyond -- against the CPU restatement of utils/sparse.py:22-165."""
    from oracle import ref_torch as R
    from torchdr_amd.utils.sparse import symmetrize_sparse

    n, k = 3000, 20
    gen = torch.Generator().manual_seed(4)
    idx = torch.empty((n, k), dtype=torch.int64)
    # column classes by how often they are drawn: in-degrees of ~2800, ~650, ~330, ~170, ~80 and ~7
    w = torch.ones(n)
    w[:4], w[4:12], w[12:20], w[20:40], w[40:100] = 900.0, 100.0, 50.0, 25.0, 12.0
    for i in range(n):
        ww = w.clone()
        ww[i] = 0                                   # no self loops
        idx[i] = torch.multinomial(ww, k, replacement=False, generator=gen)
    vals = torch.rand((n, k), generator=gen)
    V, J = symmetrize_sparse(vals.cuda(), idx.cuda())
    V2, J2 = R.symmetrize_sparse(vals, idx)
    deg = (J2 >= 0).sum(1)
    for lo, hi in ((0, 64), (65, 128), (129, 256), (257, 512), (513, 1024), (1025, 10**6)):
        assert bool(((deg >= lo) & (deg <= hi)).any()), (lo, hi, deg.max())
    assert torch.equal(J.cpu(), J2)
    assert torch.allclose(V.cpu(), V2, rtol=0, atol=1e-7)
    # a visit order of the rows (the kNN search's cluster order in the fit) changes nothing in the result
    from torchdr_amd.utils.sparse import symmetrize_to_csr

    a = symmetrize_to_csr(vals.cuda(), idx.cuda())
    b = symmetrize_to_csr(vals.cuda(), idx.cuda(), order=torch.randperm(n, generator=gen).cuda())
    assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.cols, b.cols) and torch.equal(a.vals, b.vals)


def test_symmetrize_chunked_with_ext_edges_equals_single():
    """Multi-GPU symmetrisation logic on one device: split rows into 3 'ranks', route the
    transposed edges by hand (parallel.route_edges), symmetrise each chunk -> must equal the
    corresponding rows of the single-chunk result (reference semantics of sparse.py:170-206)."""
    from torchdr_amd.distributed import chunk_bounds
    from torchdr_amd.parallel import route_edges
    from torchdr_amd.utils.sparse import symmetrize_to_csr

    a = load("affinity")
    P, I = a["umap10_P"].cuda(), a["umap10_I"].cuda()
    n = P.shape[0]
    full = symmetrize_to_csr(P, I)
    W = 3
    routed = []
    for r in range(W):
        s, e = chunk_bounds(n, r, W)
        routed.append(route_edges(P[s:e], I[s:e], s, n, W, r))
    for r in range(W):
        s, e = chunk_bounds(n, r, W)
        src = torch.cat([routed[o][r][0] for o in range(W) if o != r])
        dst = torch.cat([routed[o][r][1] for o in range(W) if o != r])
        val = torch.cat([routed[o][r][2] for o in range(W) if o != r])
        ext = ((dst - s).to(torch.int32), src, val)
        part = symmetrize_to_csr(P[s:e], I[s:e], row_offset=s, n_total=n, ext=ext)
        b0, b1 = int(full.rowptr[s]), int(full.rowptr[e])
        assert torch.equal(part.rowptr.cpu() + b0, full.rowptr[s:e + 1].cpu())
        assert torch.equal(part.cols.cpu(), full.cols[b0:b1].cpu())
        assert torch.equal(part.vals.cpu(), full.vals[b0:b1].cpu())


def test_affinity_plugins_end_to_end():
    from oracle import ref_torch as R
    from torchdr_amd.affinity import EntropicAffinity, UMAPAffinity

    a = load("affinity")
    X = a["X"].cuda()
    for nn in (10, 30):
        aff = UMAPAffinity(n_neighbors=nn, max_iter=100)
        V, J = aff(X)
        assert torch.equal(J.cpu(), a[f"umap{nn}_Isym"])
        assert torch.allclose(V.cpu(), a[f"umap{nn}_Psym"], rtol=RTOL, atol=1e-8)
        assert torch.allclose(aff.eps_.cpu(), a[f"umap{nn}_eps"], rtol=RTOL)
        P, I = UMAPAffinity(n_neighbors=nn, max_iter=100, symmetrize=False)(X)
        # the reference's row order among exactly tied distances is topk's; ours is (distance, index)
        _, Iref = R.canonical_rows(a[f"umap{nn}_C"], a[f"umap{nn}_I"])
        assert torch.equal(I.cpu(), Iref) and I.dtype == torch.int32
    for perp in (5, 30):
        aff = EntropicAffinity(perplexity=perp, max_iter=100)
        logP, I = aff(X, log=True)
        Cref, Iref = R.canonical_rows(a[f"ent{perp}_C"], a[f"ent{perp}_I"])
        assert torch.equal(I.cpu(), Iref)
        order = torch.argsort(a[f"ent{perp}_C"], dim=1, stable=True)  # logP is monotone in C within a row
        ref = torch.gather(a[f"ent{perp}_logP"], 1, order)
        assert torch.allclose(logP.cpu(), ref, rtol=RTOL, atol=1e-4)
        Pn, _ = aff(X)  # log=False exponentiates (affinity/base.py:554)
        assert torch.allclose(Pn.sum(1).cpu(), torch.full((X.shape[0],), 1.0 / X.shape[0]), rtol=1e-4)
    # numpy input is accepted by the plugin surface
    V2, J2 = UMAPAffinity(n_neighbors=10, max_iter=100)(a["X"].numpy())
    assert torch.equal(J2.cpu(), a["umap10_Isym"])


def test_dense_affinities_sparsity_false():
    """sparsity=False: full N x N rows through the streaming search kernels, vs the real reference."""
    from torchdr_amd.affinity import EntropicAffinity, UMAPAffinity

    g = load("affinity_dense")
    X = g["X"].cuda()
    n = X.shape[0]
    aff = EntropicAffinity(perplexity=10, sparsity=False, max_iter=100)
    logP = aff(X, log=True, return_indices=False)
    assert logP.shape == (n, n)
    assert torch.allclose(aff.eps_.cpu(), g["ent_eps"], rtol=RTOL)
    off = ~torch.eye(n, dtype=torch.bool)
    assert torch.allclose(logP.cpu()[off], g["ent_logP"][off], rtol=RTOL, atol=1e-4)
    assert torch.isinf(logP.cpu().diagonal()).all() or (logP.cpu().diagonal() < -1e6).all()  # zero diagonal
    P, idx = UMAPAffinity(n_neighbors=10, sparsity=False, max_iter=100)(X)
    assert idx is None and P.shape == (n, n)
    assert torch.allclose(P.cpu(), g["umap_P"], rtol=1e-4, atol=1e-7)
    assert torch.allclose(P, P.T)


def test_symmetrisation_of_blocks_wider_than_the_row_local_kernels():
    """k > 256 (the kNN stage serves up to 1024 neighbours; `UMAP(n_neighbors=300)`): the sort-and-coalesce form of
    utils/sparse.py.  On a block the kernels DO take (k = 40) it must give the kernels' CSR -- same pattern, values to fp32
    rounding -- for both modes; a 300-wide block against a dense evaluation; UMAP with 300 neighbours end to end."""
    import torchdr_amd
    from torchdr_amd.utils.sparse import _symmetrize_wide, symmetrize_to_csr

    gen = torch.Generator().manual_seed(5)
    n, k = 900, 40
    I = torch.stack([torch.randperm(n, generator=gen)[:k] for _ in range(n)]).cuda()
    P = torch.rand(n, k, generator=gen).cuda()
    for mode in ("sum_minus_prod", "sum"):
        a = symmetrize_to_csr(P, I, mode)
        b = _symmetrize_wide(P, I, mode, 0, n, None)
        assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.cols, b.cols)
        assert torch.allclose(a.vals, b.vals, rtol=1e-6, atol=1e-7)
    n, k = 700, 300
    I = torch.stack([torch.randperm(n, generator=gen)[:k] for _ in range(n)]).cuda()
    P = torch.rand(n, k, generator=gen).cuda()
    c = symmetrize_to_csr(P, I, "sum_minus_prod")
    D = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    D[torch.arange(n, device="cuda")[:, None].expand(n, k), I] = P.double()
    Q = D + D.T - D * D.T
    got = torch.zeros_like(Q)
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), c.rowptr[1:] - c.rowptr[:-1])
    got[rows, c.cols.long()] = c.vals.double()
    assert torch.equal(got != 0, Q != 0) and torch.allclose(got, Q, rtol=1e-6, atol=1e-7)
    X = gmm(1500, 16, 2.0, seed=8).cuda()
    Z = torchdr_amd.UMAP(n_neighbors=300, max_iter=40, random_state=0).fit_transform(X)
    assert Z.shape == (1500, 2) and bool(torch.isfinite(Z).all())
