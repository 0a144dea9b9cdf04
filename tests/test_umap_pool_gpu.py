"""Pool-sampled UMAP gradient (csrc/tdr_umap_pool.hip) -- the kernel bench.py times since round 6.

* the gradient with ITS OWN in-kernel negatives against the oracle (oracle.ref_torch.umap_gradients = umap.py:236-292) evaluated
  on the negatives the kernel draws, dumped by tdr_umap_pool_debug_negatives (same device functions), at 1e-5 of max |g|;
* the sampler's law: uniform marginal (chi-square over rows), dropped draws exactly where the header says (self, padding of the
  last run), pools of different iterations / blocks independent;
* a row-sharded launch gives the bits of the full launch (a row's sums are a function of the row alone);
* the estimator: embedding quality against the REFERENCE's own scores (tests/golden/quality2.json) on four data regimes and two
  sizes, next to the i.i.d. sampler's, and the reference's silhouette check (tests/test_neighbor_embedding.py:42-74).
"""

import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import gmm, regime_data
from tests.test_oracle_golden import load
from tests.test_umap_sched_gpu import Sched, csr_rows_to_padded, padded_to_csr, prepare, random_graph

pytestmark = pytest.mark.gpu

# geometry -> (rows per block, pool runs, rows per run): TDR_POOL_GEOMS of csrc/tdr_umap_pool.hip; 0 = the production default
GEOMS = {0: (1024, 256, 8), 1: (512, 256, 16), 2: (1024, 256, 16), 3: (1024, 128, 16), 4: (512, 256, 8), 5: (1024, 512, 8), 6: (1024, 256, 16)}


def pool_grad(sc, Z, t_local, n_iter, a, b, n_neg, seed, geom=0, neg_rate=5):
    from torchdr_amd import _lib

    g = torch.full((sc.n_rows, sc.nc), float("nan"), device="cuda")
    _lib.check(_lib.lib().tdr_umap_pool_grad_f32(_lib.ptr(Z), sc.nc, sc.n_total, sc.row0, sc.n_rows, _lib.ptr(sc.list), _lib.ptr(sc.hdr),
                                                 t_local, a, b, n_iter, neg_rate, n_neg, seed, 1.0, 1.0, 1e-3, _lib.ptr(g), geom,
                                                 _lib.stream_ptr()), "pool_grad")
    return g


def pool_negatives(seed, n_iter, n_total, row0, nuse, geom, width):
    from torchdr_amd import _lib

    out = torch.empty((nuse.numel(), width), dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().tdr_umap_pool_debug_negatives(seed, n_iter, n_total, row0, nuse.numel(), _lib.ptr(nuse), geom, width,
                                                        _lib.ptr(out), _lib.stream_ptr()), "pool_debug_negatives")
    return out


def oracle_check_pool(sc, Z, nxt_before, t_local, n_iter, a, b, n_neg, seed, rows, geom=0):
    """Max error of the pool kernel's gradient on `rows` relative to max |g|, against the oracle on the kernel's own draws."""
    import oracle.ref_torch as R

    grad = pool_grad(sc, Z, t_local, n_iter, a, b, n_neg, seed, geom=geom).cpu()
    assert bool(torch.isfinite(grad).all())
    n = sc.n_rows
    act = sc.records(sc.B)[2][t_local].to(torch.int32).cuda()
    nuse = torch.clamp(act * 5, max=n_neg).to(torch.int32).contiguous()
    width = max(int(nuse.max()), 1)
    neg = pool_negatives(seed, n_iter, sc.n_total, sc.row0, nuse, geom, width).cpu()[rows]
    assert torch.equal((neg != -1).sum(1), nuse.cpu().long()[rows])
    # dropped draws (-2) and unused slots (-1): the row itself -- a zero difference contributes nothing in the oracle either
    own = (rows + sc.row0)[:, None].expand_as(neg)
    neg = torch.where(neg >= 0, neg, own)
    if width < n_neg:
        neg = torch.cat([neg, own[:, :1].expand(-1, n_neg - width)], 1)
    rowptr, cols = sc.rowptr.cpu(), sc.cols.cpu()
    NN, (ep_p, nx_p) = csr_rows_to_padded(rowptr, cols, [sc.eps_per.cpu(), nxt_before.cpu()], rows, [float("inf"), float("inf")])
    ga, gr, act_o = R.umap_gradients(Z.cpu(), NN, ep_p, nx_p, neg, n_iter, a, b, rows=rows + sc.row0)
    assert torch.equal(act_o.sum(1), act.cpu().long()[rows])
    ref = ga + gr
    return float((grad[rows] - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("nc", [2, 3])
def test_pool_gradient_vs_oracle_fixture_graph(nc):
    """The production path on the reference's own affinity graph of the umap_step fixture: every geometry, three window
    positions, n not a multiple of 16 by construction of the second case (rows dropped from the end)."""
    g = load("umap_step")
    a, b, T = float(g["a"]), float(g["b"]), int(g["max_iter"])
    n_full = g["X"].shape[0]
    for n in (n_full, n_full - 5):
        P, I = g["Psym"][:n].clone(), g["Isym"][:n].clone()
        I[I >= n] = -1
        rowptr, cols, vals = (t.cuda() for t in padded_to_csr(P, I))
        eps_per, nxt = prepare(vals, T)
        sc = Sched(rowptr, cols, eps_per, n, 32, 1, nc=nc)
        gen = torch.Generator().manual_seed(11)
        Z = (torch.randn(n, nc, generator=gen) * 3).cuda().contiguous()
        rows = torch.arange(n)
        before = nxt.clone()
        sc.build(nxt, 0, 32)
        ep = eps_per.cpu()
        for tl in (0, 13, 31):
            nb = before.cpu().clone()
            for t in range(tl):
                act = nb <= np.float32(t + 1)
                nb[act] += ep[act]
            for geom in GEOMS:
                err = oracle_check_pool(sc, Z, nb, tl, tl, a, b, 50, 1234567, rows, geom=geom)
                assert err < 1e-5, (nc, n, tl, geom, err)


def test_pool_gradient_vs_oracle_large_and_wide():
    """N = 500 003 (ragged last run) with n_neighbors = 60 (300 negatives per row at most); oracle on 4096 sampled rows."""
    from torchdr_amd.affinity import UMAPAffinity

    n = 500_003
    X = gmm(n, 32, 2.0, seed=7).cuda()
    csr = UMAPAffinity(n_neighbors=60, max_iter=100)(X, return_csr=True)
    del X
    eps_per, nxt = prepare(csr.vals, 500)
    sc = Sched(csr.rowptr, csr.cols, eps_per, n, 32, 1)
    for t0 in (0, 32, 64):
        before = nxt.clone()
        sc.build(nxt, t0, 32)
    gen = torch.Generator().manual_seed(1)
    Z = (torch.randn(n, 2, generator=gen) * 4).cuda().contiguous()
    rows = torch.cat([torch.randperm(n, generator=gen)[:4090], torch.arange(n - 6, n)]).sort().values
    err = oracle_check_pool(sc, Z, before, 0, 64, 1.577, 0.895, 300, 99, rows)
    assert err < 1e-5, err


@pytest.mark.parametrize("geom", [0, 1, 4, 6])
def test_pool_row_shard_gives_the_bits_of_the_full_launch(geom):
    """Chunks that start and end inside a row block: the rows' gradients equal the full launch bit for bit (the pool is keyed by
    the GLOBAL row block, a row's sums by the row alone)."""
    n = 7013
    rowptr, cols, vals = random_graph(n, seed=77)
    eps_per, nxt = prepare(vals.cuda(), 200)
    sc = Sched(rowptr.cuda(), cols.cuda(), eps_per, n, 32, 1)
    before = nxt.clone()
    sc.build(nxt, 0, 32)
    gen = torch.Generator().manual_seed(5)
    Z = (torch.randn(n, 2, generator=gen) * 3).cuda().contiguous()
    full = pool_grad(sc, Z, 9, 9, 1.577, 0.895, 40, 4242, geom=geom)
    again = pool_grad(sc, Z, 9, 9, 1.577, 0.895, 40, 4242, geom=geom)
    assert torch.equal(full, again)
    if geom == 0:
        # a block worked by 1, 2, 4 or 8 workgroups (what the launcher picks by launch size: small shards fill the chip): same bits
        for g2 in (17, 18, 20, 24):
            assert torch.equal(pool_grad(sc, Z, 9, 9, 1.577, 0.895, 40, 4242, geom=g2), full), g2
    for c0, c1 in ((0, 2338), (2338, 4676), (4676, n), (1000, 1001), (300, 1500)):
        rp = (rowptr[c0:c1 + 1] - rowptr[c0]).cuda()
        e0, e1 = int(rowptr[c0]), int(rowptr[c1])
        sub = Sched(rp, cols[e0:e1].cuda(), eps_per[e0:e1].contiguous(), n, 32, 1, row0=c0)
        nx = before[e0:e1].clone()
        sub.build(nx, 0, 32)
        part = pool_grad(sub, Z, 9, 9, 1.577, 0.895, 40, 4242, geom=geom)
        assert torch.equal(part, full[c0:c1]), (geom, c0, c1)


@pytest.mark.parametrize("with_alt,T", [(True, 71), (True, 70), (False, 71)])
@pytest.mark.parametrize("use_graph", [0, 1])
def test_loop_object_with_the_step_in_the_pool_launch(use_graph, with_alt, T):
    """tdr_umap_loop_run on pool negatives against the same windows issued call by call (build, tdr_umap_pool_grad_f32,
    tdr_sgd_step_f32): with a second embedding buffer the gradient launch carries the step (learning rate from the device table,
    gradient / norm / snapshot at the inspected iterations) and the buffers swap roles every iteration -- bit-identical embedding
    after 70 / 71 iterations (windows 32 + 32 + 6 or 7: an odd window copies the rows back), same snapshot and norms; without
    the second buffer the unfused sequence runs as before."""
    import ctypes

    from torchdr_amd import _lib
    from tests.test_umap_sched_gpu import layout

    L = _lib.lib()
    n, ci = 20000, 25
    gen = torch.Generator().manual_seed(3)
    rowptr, cols, vals = random_graph(n, seed=9, hub=300)
    eps_csr, _ = prepare(vals.cuda(), 200)
    cols_p, eps_p = layout(rowptr.cuda(), cols.cuda(), eps_csr)
    rowptr = rowptr.cuda()
    lr = torch.linspace(1.0, 0.0, T + 1)[:T].contiguous()
    Z0 = (torch.randn(n, 2, generator=gen) * 3).cuda()
    a, b, seed = 1.577, 0.895, 4242
    sc = Sched(rowptr, cols_p, eps_p, n, 32, 1)
    Z, nxt = Z0.clone(), eps_p.clone()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    norms, snaps = {}, {}
    for t0 in range(0, T, 32):
        nw = min(32, T - t0)
        sc.build(nxt, t0, nw)
        for tl in range(nw):
            g = pool_grad(sc, Z, tl, t0 + tl, a, b, 150, seed)
            if (t0 + tl) % ci == 0:
                norms[(t0 + tl) // ci] = float((g.double() ** 2).sum())
            _lib.check(L.tdr_sgd_step_f32(_lib.ptr(Z), _lib.ptr(g), None, Z.numel(), float(lr[t0 + tl]), 0.0,
                                          1 if t0 + tl == 0 else 0, _lib.ptr(flag), t0 + tl, _lib.stream_ptr()), "sgd")
            if (t0 + tl) % ci == 0:
                snaps[t0 + tl] = Z.clone()
    sc2 = Sched(rowptr, cols_p, eps_p, n, 32, 1)
    Z2, nxt2 = Z0.clone(), eps_p.clone()
    Zalt = torch.full_like(Z0, float("nan"))
    grad = torch.empty((n, 2), device="cuda")
    lr_d, norm2 = lr.cuda(), torch.zeros(T // ci + 2, device="cuda")
    flag2, scratch = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(16, dtype=torch.int32, device="cuda")
    snap = torch.zeros((n, 2), device="cuda")
    d = _lib.UmapLoopDesc()
    d.Z, d.nc, d.n_total, d.row0, d.n_rows = _lib.ptr(Z2), 2, n, 0, n
    d.rowptr, d.cols, d.eps_per, d.next = _lib.ptr(rowptr), _lib.ptr(cols_p), _lib.ptr(eps_p), _lib.ptr(nxt2)
    d.blk_base, d.list, d.hdr, d.err = _lib.ptr(sc2.blk_base), _lib.ptr(sc2.list), _lib.ptr(sc2.hdr), _lib.ptr(sc2.err)
    d.acc, d.grad, d.mom_buf = None, _lib.ptr(grad), None
    d.a, d.b, d.neg_rate, d.n_negatives, d.seed = a, b, 5, 150, seed
    d.exag, d.rep, d.eps, d.n_slices, d.block_iters = 1.0, 1.0, 1e-3, 1, 32
    d.lr_table, d.max_iter, d.momentum, d.first_iter, d.check_interval = _lib.ptr(lr_d), T, 0.0, 0, ci
    d.norm2, d.snap, d.nan_flag, d.scratch, d.gather, d.gather_ctx, d.geom = (_lib.ptr(norm2), _lib.ptr(snap), _lib.ptr(flag2), _lib.ptr(scratch),
                                                                               None, None, 0)
    d.pool = 1
    d.Z_alt = _lib.ptr(Zalt) if with_alt else None
    h = ctypes.c_void_p()
    _lib.check(L.tdr_umap_loop_create(ctypes.byref(h), ctypes.byref(d)), "create")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(side):
            _lib.check(L.tdr_umap_loop_run(h, 0, 64, use_graph, _lib.stream_ptr()), "run")
            _lib.check(L.tdr_umap_loop_run(h, 64, T - 64, use_graph, _lib.stream_ptr()), "run")
        torch.cuda.synchronize()
        assert torch.equal(Z2, Z) and torch.equal(nxt2, nxt)
        assert torch.equal(snap, snaps[(T - 1) // ci * ci])
        for k, v in norms.items():
            assert abs(float(norm2[k]) - v) <= 1e-5 * v
    finally:
        L.tdr_umap_loop_destroy(h)
    assert int(sc2.err.item()) == 0 and int(flag2.item()) == 0
    # the second buffer must be a different, 16-byte aligned one
    d.Z_alt = _lib.ptr(Z2)
    assert L.tdr_umap_loop_create(ctypes.byref(h), ctypes.byref(d)) != 0


@pytest.mark.parametrize("geom", [0, 1, 5])
def test_pool_sampler_law(geom):
    """Marginal law of an item: uniform over the rows (chi-square on 5003 rows, ragged last run), dropped draws = the row
    itself or the padding rows of the last run at the expected rate; pools are redrawn per iteration and per block
    (the runs staged by two iterations / two blocks overlap as independent uniform samples do)."""
    n = 5003
    rows_per_block, runs, rl = GEOMS[geom]
    n_runs = (n + rl - 1) // rl
    p_drop = (rl * n_runs - n + 1) / (rl * n_runs)   # padding rows (rl n_runs - n of rl n_runs slots) + self (1 slot)
    for width, iters in ((2, 200), (150, 30)):
        nuse = torch.full((n,), width, dtype=torch.int32, device="cuda")
        counts = torch.zeros(n, dtype=torch.float64)
        dropped = total = 0
        pools = []
        for it in range(iters):
            neg = pool_negatives(987654321, it, n, 0, nuse, geom, width).cpu()
            assert int((neg == -1).sum()) == 0
            keep = neg >= 0
            assert not bool((neg == torch.arange(n)[:, None]).any()) and int(neg.max()) < n
            counts += torch.bincount(neg[keep], minlength=n).double()
            dropped += int((neg == -2).sum())
            total += neg.numel()
            if width > 2:   # the pool a block used this iteration ~ the set of runs its rows drew from
                blk = [neg[b * rows_per_block:(b + 1) * rows_per_block] for b in range(2)]
                pools.append([set((x[x >= 0] // rl).tolist()) for x in blk])
        # rows of a block share the iteration's pool (and the 16 rows of a run enter it together): counts are over-dispersed
        # against i.i.d. draws by 1 + (draws per pool row) -- the drop rate by the same factor
        infl = 1.0 + rows_per_block * width / float(rl * runs)
        assert abs(dropped / total - p_drop) < 6 * np.sqrt(p_drop * infl * rl / total) + 1e-4, (width, dropped / total, p_drop)
        kept = counts.sum()
        chi2 = float(((counts - kept / n) ** 2 / (kept / n)).sum())
        assert 0.6 * (n - 1) * infl < chi2 + (n - 1) * (infl - 1) * 0.6 and chi2 < 1.5 * (n - 1) * infl, (width, chi2, n, infl)
        assert int((counts == 0).sum()) == 0     # every row is drawn
        if width == 2:
            # with few draws per pool row the counts are close to i.i.d.: a real chi-square test of the uniform marginal
            assert chi2 < (n - 1) * infl + 24 * np.sqrt(2 * (n - 1)) * infl, (chi2, n, infl)
        # a pool holds at most `runs` runs; pools of consecutive iterations / of two blocks are different samples
        for it in range(1, len(pools)):
            a_, b_, c_ = pools[it][0], pools[it - 1][0], pools[it][1]
            assert len(a_) <= runs and len(c_) <= runs
            assert a_ != b_ and a_ != c_


def _scores(X, Z, lab):
    from sklearn.metrics import silhouette_score

    from torchdr_amd.eval import knn_label_accuracy, neighborhood_preservation

    return {"np": float(neighborhood_preservation(X, Z, K=15)), "acc": float(knn_label_accuracy(Z, lab.cuda(), k=10)),
            "sil": float(silhouette_score(Z.cpu().numpy(), lab.numpy(), sample_size=5000, random_state=0))}


QUALITY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quality2.json")


@pytest.mark.parametrize("regime", ["gmm2", "overlap", "swiss", "heavytail"])
@pytest.mark.parametrize("n", [5000, 20000])
def test_pool_sampler_gives_the_reference_quality(regime, n):
    """The same fit (500 iterations) as the REFERENCE's run recorded in tests/golden/quality2.json, with the pool sampler and
    with the i.i.d. one: neighbourhood preservation, 10-NN label accuracy and silhouette of the embedding.  Both samplers must
    reach the reference's worst seed up to its own seed-to-seed spread (floor: 10 % / 0.02 / 0.05)."""
    import torchdr_amd
    from torchdr_amd import config

    q = [c for c in json.load(open(QUALITY))["cases"] if c["regime"] == regime and c["n"] == n]
    assert len(q) >= 2
    X, lab = regime_data(regime, n)
    Xc = X.cuda()
    out = {}
    geom = int(os.environ.get("TDR_TEST_POOL_GEOM", "0"))     # measurement runs: another pool geometry than the default
    for mode in ("pool", "iid"):
        with config.options(NEGATIVES=mode, POOL_GEOM=geom):
            s = [_scores(Xc, torchdr_amd.UMAP(n_neighbors=30, max_iter=500, random_state=seed).fit_transform(Xc), lab) for seed in (0, 1)]
        out[mode] = {k: min(r[k] for r in s) for k in s[0]}
    ref = {"np": [c["neighborhood_preservation_K15"] for c in q], "acc": [c["knn_label_accuracy_k10"] for c in q],
           "sil": [c["silhouette"] for c in q]}
    rec = {"regime": regime, "n": n, "pool_geom": geom, "reference": {k: min(v) for k, v in ref.items()}, **out}
    print(rec)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/pool_quality.jsonl", "a") as f:
        f.write(json.dumps(rec) + "\n")
    for mode in ("pool", "iid"):
        got = out[mode]
        assert got["np"] >= min(ref["np"]) - max(0.1 * min(ref["np"]), max(ref["np"]) - min(ref["np"])), (mode, got, ref)
        assert got["acc"] >= min(ref["acc"]) - max(0.02, max(ref["acc"]) - min(ref["acc"])), (mode, got, ref)
        assert got["sil"] >= min(ref["sil"]) - max(0.05, max(ref["sil"]) - min(ref["sil"])), (mode, got, ref)


def test_pool_estimator_is_reproducible_and_falls_back_for_injected_negatives():
    """Two fits with the same random_state agree bit for bit; a fit whose negatives are injected (the reference's
    `neg_indices_`, parity tests) runs the i.i.d. kernel on the same one-slice lists."""
    import torchdr_amd

    X = gmm(6000, 32, 2.0, seed=3).cuda()
    Z1 = torchdr_amd.UMAP(n_neighbors=15, max_iter=120, random_state=5).fit_transform(X)
    Z2 = torchdr_amd.UMAP(n_neighbors=15, max_iter=120, random_state=5).fit_transform(X)
    assert torch.equal(Z1, Z2)

    class Injected(torchdr_amd.UMAP):
        def on_training_step_start(self):
            super().on_training_step_start()
            gen = torch.Generator(device="cuda").manual_seed(int(self.n_iter_))
            r = torch.randint(0, 5999, (6000, self.n_negatives), device="cuda", generator=gen)
            self.neg_indices_ = r + (r >= torch.arange(6000, device="cuda")[:, None]).long()

    Z3 = Injected(n_neighbors=15, max_iter=60, random_state=5).fit_transform(X)
    assert bool(torch.isfinite(Z3).all())
