"""Scheduled UMAP loop (csrc/tdr_umap_sched.hip) -- the kernels bench.py times.

* the schedule kernel against a step-by-step restatement of umap.py:243-247 (bit-exact counters, exact lists);
* the gradient kernel with the reference's own negatives against the golden vectors of the real reference;
* the gradient kernel with ITS OWN in-kernel negatives against the oracle: the negatives it draws are dumped by
  tdr_umap_debug_negatives (same device functions) and fed to oracle.ref_torch.umap_gradients.
"""

import numpy as np
import pytest
import torch

from tests.conftest import gmm
from tests.test_oracle_golden import load

pytestmark = pytest.mark.gpu


def padded_to_csr(V, J):
    mask = J >= 0
    rowptr = torch.zeros(V.shape[0] + 1, dtype=torch.int64)
    rowptr[1:] = mask.sum(1).cumsum(0)
    return rowptr, J[mask].to(torch.int32), V[mask]


class Sched:
    """Thin ctypes driver of plan / build / grad for the tests."""

    def __init__(self, rowptr, cols, eps_per, n_total, B, S, nc=2, row0=0):
        from torchdr_amd import _lib

        self.lib, self.L = _lib, _lib.lib()
        self.rowptr, self.cols, self.eps_per = rowptr, cols, eps_per
        self.n_rows, self.n_total, self.B, self.S, self.nc, self.row0 = rowptr.numel() - 1, n_total, B, S, nc, row0
        nb = (self.n_rows + 63) // 64
        self.nb = nb
        scratch = torch.empty(nb, dtype=torch.int64, device="cuda")
        self.blk_base = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
        _lib.check(self.L.tdr_umap_sched_plan_f32(_lib.ptr(rowptr), _lib.ptr(eps_per), self.n_rows, B, _lib.ptr(scratch),
                                                  _lib.ptr(self.blk_base), _lib.stream_ptr()), "plan")
        cap = int(self.blk_base[-1].item())
        self.list = torch.full((cap + 64,), -7, dtype=torch.int32, device="cuda")  # + slack read by idle lanes
        self.hdr = torch.zeros((int(self.L.tdr_umap_sched_hdr_entries(self.n_rows, B, S)), 2), dtype=torch.int32, device="cuda")
        self.err = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.acc = torch.empty((S * self.n_rows, 2 * nc), device="cuda")   # S planes: enough for the joint launch (geom & 16)

    def build(self, nxt, t0, n):
        _l = self.lib
        _l.check(self.L.tdr_umap_sched_build_f32(_l.ptr(self.rowptr), _l.ptr(self.cols), _l.ptr(self.eps_per), _l.ptr(nxt),
                                                 self.n_rows, self.n_total, t0, n, self.S, _l.ptr(self.blk_base),
                                                 _l.ptr(self.list), _l.ptr(self.hdr), _l.ptr(self.err), _l.stream_ptr()), "build")
        assert int(self.err.item()) == 0

    def grad(self, Z, t_local, n_iter, a, b, n_neg, neg=None, seed=0, geom=0, neg_rate=5):
        _l = self.lib
        g = torch.empty((self.n_rows, self.nc), device="cuda")
        _l.check(self.L.tdr_umap_sched_grad_f32(_l.ptr(Z), self.nc, self.n_total, self.row0, self.n_rows, _l.ptr(self.list),
                                                _l.ptr(self.hdr), t_local, self.S, a, b, n_iter, neg_rate, n_neg, _l.ptr(neg), seed,
                                                1.0, 1.0, 1e-3, _l.ptr(g), _l.ptr(self.acc), geom, _l.stream_ptr()), "sched_grad")
        return g

    def records(self, n_iters):
        """(start, length, act) of every (iteration, slice, row) segment, as int64 CPU tensors of shape (n_iters*S, n_rows)."""
        h = self.hdr.cpu().long().view(-1, self.n_rows, 2)[: n_iters * self.S]
        start = h[..., 0] & 0xFFFFFFFF
        return start, h[..., 1] & 0xFFFF, (h[..., 1] >> 16) & 0xFFFF


def random_graph(n, seed, hub=700):
    gen = torch.Generator().manual_seed(seed)
    deg = torch.randint(0, 90, (n,), generator=gen)
    deg[5] = hub          # a hub row: many 16-edge chunks in one row group
    deg[64:70] = 0        # empty rows at a block boundary
    deg[n - 1] = 33
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = deg.cumsum(0)
    nnz = int(rowptr[-1])
    cols = torch.randint(0, n, (nnz,), generator=gen, dtype=torch.int32)
    vals = torch.rand(nnz, generator=gen) ** 3     # many small weights -> long periods, some below the cut (inf)
    vals[torch.rand(nnz, generator=gen) < 0.05] = 1.0
    return rowptr, cols, vals


def prepare(vals, max_iter):
    from torchdr_amd import _lib

    nnz = vals.numel()
    eps_per = torch.empty(nnz, device="cuda")
    nxt = torch.empty(nnz, device="cuda")
    scratch = torch.zeros(2, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().tdr_umap_prepare_f32(_lib.ptr(vals), nnz, max_iter, _lib.ptr(eps_per), _lib.ptr(nxt), _lib.ptr(scratch),
                                               _lib.stream_ptr()), "prepare")
    return eps_per, nxt


def layout(rowptr, cols, eps_per):
    from torchdr_amd import _lib

    cols_p, eps_p = torch.empty_like(cols), torch.empty_like(eps_per)
    _lib.check(_lib.lib().tdr_umap_sched_layout_f32(_lib.ptr(rowptr), _lib.ptr(cols), _lib.ptr(eps_per), rowptr.numel() - 1,
                                                    _lib.ptr(cols_p), _lib.ptr(eps_p), _lib.stream_ptr()), "layout")
    return cols_p, eps_p


def test_loop_layout_sorts_every_row_by_firing_period():
    """tdr_umap_sched_layout_f32: each row's (column, epochs_per_sample) pairs, sorted by (epochs_per_sample, column)."""
    n = 3000
    rowptr, cols, vals = random_graph(n, seed=21, hub=2500)   # the hub row exceeds the sorted range: left as it is
    eps_per, _ = prepare(vals.cuda(), 200)
    cols_p, eps_p = layout(rowptr.cuda(), cols.cuda(), eps_per)
    ep, cp, epp = eps_per.cpu(), cols_p.cpu(), eps_p.cpu()
    for r in list(range(0, 80)) + [n - 1]:
        e0, e1 = int(rowptr[r]), int(rowptr[r + 1])
        if e1 - e0 > 2048:
            assert torch.equal(cp[e0:e1], cols[e0:e1]) and torch.equal(epp[e0:e1], ep[e0:e1])
            continue
        # by (epochs_per_sample, column, position)
        by_col = torch.sort(cols[e0:e1], stable=True).indices
        order = by_col[torch.sort(ep[e0:e1][by_col], stable=True).indices]
        assert torch.equal(epp[e0:e1], ep[e0:e1][order])
        assert torch.equal(cp[e0:e1], cols[e0:e1][order])


def test_loop_layout_rows_of_every_length_class():
    """Rows of 1 .. 64 edges (one entry per lane), 65 .. 1024 (2 / 4 / 8 / 16 entries per lane in registers) and beyond (general
    path), with tied periods and -- in a second pass -- a negative period that sends its row to the general comparison: every row
    sorted by (epochs_per_sample, column)."""
    gen = torch.Generator().manual_seed(8)
    degs = [1, 2, 63, 64, 65, 100, 128, 129, 200, 256, 257, 400, 512, 513, 900, 1024, 1025, 1500, 0, 33]
    n = len(degs)
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.tensor(degs).cumsum(0)
    nnz = int(rowptr[-1])
    cols = torch.cat([torch.randperm(5000, generator=gen)[:d] for d in degs]).to(torch.int32)
    eps = torch.randint(1, 40, (nnz,), generator=gen).float() * 0.5        # many ties: the column decides
    eps[torch.rand(nnz, generator=gen) < 0.1] = float("inf")
    for neg in (False, True):
        ep = eps.clone()
        if neg:
            ep[rowptr[8] + 5] = -1.0      # row of 200 edges
            ep[rowptr[2] + 7] = -2.0      # row of 63 edges
        cols_p, eps_p = layout(rowptr.cuda(), cols.cuda(), ep.cuda())
        cp, epp = cols_p.cpu(), eps_p.cpu()
        for r in range(n):
            e0, e1 = int(rowptr[r]), int(rowptr[r + 1])
            by_col = torch.sort(cols[e0:e1], stable=True).indices
            order = by_col[torch.sort(ep[e0:e1][by_col], stable=True).indices]
            assert torch.equal(epp[e0:e1], ep[e0:e1][order]), (neg, r, degs[r])
            assert torch.equal(cp[e0:e1], cols[e0:e1][order]), (neg, r, degs[r])


@pytest.mark.parametrize("S", [1, 2, 4, 8])
@pytest.mark.parametrize("B,t0", [(32, 0), (7, 37)])
def test_schedule_is_the_step_by_step_recurrence(S, B, t0):
    """umap.py:243-247 iterated B times on the CPU (fp32 compare and add, one firing per iteration at most) gives the
    counters after the window and, per iteration, the set of firing edges: the kernel's `next`, `act` table and the
    content of every (iteration, slice, row) list segment must equal them exactly (the order inside a segment only
    permutes the force sum; it must be the same on every run)."""
    n = 3000
    rowptr, cols, vals = random_graph(n, seed=3 + S)
    eps_per, nxt = prepare(vals.cuda(), 200)
    ep_c, nx_c = eps_per.cpu(), nxt.cpu().clone()
    # bring the counters to iteration t0 with the reference recurrence
    for t in range(t0):
        a = nx_c <= np.float32(t + 1)
        nx_c[a] += ep_c[a]
    nx0 = nx_c.clone()
    nxt = nx_c.clone().cuda()
    sc = Sched(rowptr.cuda(), cols.cuda(), eps_per, n, B, S)
    sc.build(nxt, t0, B)
    fires = []
    for t in range(t0, t0 + B):
        a = nx_c <= np.float32(t + 1)
        nx_c[a] += ep_c[a]
        fires.append(a)
    assert torch.equal(nxt.cpu(), nx_c)
    row_of = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
    step = (n - 1 + S - 1) // S
    sl_of = torch.clamp(cols.long() // step, max=S - 1)
    start, length, act = sc.records(B)
    lst, base = sc.list.cpu().long(), sc.blk_base.cpu()
    for t in range(B):
        want_act = torch.bincount(row_of[fires[t]], minlength=n)
        for s in range(S):
            k = t * S + s
            assert torch.equal(act[k], want_act)
            sel = fires[t] & (sl_of == s)
            assert torch.equal(length[k], torch.bincount(row_of[sel], minlength=n))
            # segment content (sorted: content, not order), every segment read through its own record
            idx = torch.repeat_interleave(start[k], length[k]) + (torch.arange(int(length[k].sum())) -
                                                                   torch.repeat_interleave(length[k].cumsum(0) - length[k], length[k]))
            rows_g = torch.repeat_interleave(torch.arange(n), length[k])
            assert torch.equal(torch.sort(row_of[sel] * n + cols[sel].long()).values, torch.sort(rows_g * n + lst[idx]).values)
    # same lists on every run
    lst_first = sc.list.clone()
    nxt2 = nx0.clone().cuda()
    sc.build(nxt2, t0, B)
    assert torch.equal(sc.list, lst_first) and torch.equal(nxt2, nxt)
    # segments tile each block's region without gaps, in (iteration, slice, row) order
    nb = sc.nb
    pad = nb * 64 - n
    st = torch.cat([start, start[:, -1:].expand(-1, pad) + length[:, -1:].expand(-1, pad)], 1).view(B * S, nb, 64)
    ln = torch.cat([length, torch.zeros((B * S, pad), dtype=torch.long)], 1).view(B * S, nb, 64)
    assert torch.equal(st[:, :, 1:], st[:, :, :-1] + ln[:, :, :-1])           # rows follow each other inside a segment run
    assert torch.equal(st[0, :, 0], base[:-1])                                 # a block's first segment opens its region
    assert torch.equal(st[1:, :, 0], st[:-1, :, 63] + ln[:-1, :, 63])         # (iteration, slice) runs follow each other
    assert bool((st[-1, :, 63] + ln[-1, :, 63] <= base[1:]).all())            # and stay inside the region


@pytest.mark.parametrize("S", [1, 2, 4])
def test_scheduled_gradient_three_steps_vs_reference(S):
    """The reference's own three UMAP steps (tests/golden/umap_step.npz: embedding, counters and the negatives it drew):
    scheduled kernels with injected negatives reproduce its gradient (1e-5) and its counters (bit-exact)."""
    g = load("umap_step")
    a, b, T = float(g["a"]), float(g["b"]), int(g["max_iter"])
    rowptr, cols, vals = (t.cuda() for t in padded_to_csr(g["Psym"], g["Isym"]))
    n = g["X"].shape[0]
    eps_per, _ = prepare(vals, T)
    mask = g["Isym"] >= 0
    sc = Sched(rowptr, cols, eps_per, n, 32, S)
    for t in range(3):
        Z = g[f"Z_{t}"].cuda().contiguous()
        nxt = g[f"next_{t}"][mask].cuda().contiguous()
        neg = g[f"neg_{t}"].cuda().contiguous()
        sc.build(nxt, t, 1)
        grad = sc.grad(Z, 0, t, a, b, neg.shape[1], neg=neg)
        ref = g[f"grad_{t}"]
        assert torch.allclose(grad.cpu(), ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
        assert torch.equal(nxt.cpu(), g[f"nextafter_{t}"][mask])


def csr_rows_to_padded(rowptr, cols, per_edge, rows, fill):
    """Padded (len(rows), max_deg) views of the CSR rows `rows` (CPU tensors)."""
    deg = (rowptr[rows + 1] - rowptr[rows])
    K = int(deg.max())
    NN = torch.zeros((rows.numel(), K), dtype=torch.int64)
    out = [torch.full((rows.numel(), K), f, dtype=p.dtype) for p, f in zip(per_edge, fill)]
    for i, r in enumerate(rows.tolist()):
        e0, e1 = int(rowptr[r]), int(rowptr[r + 1])
        NN[i, : e1 - e0] = cols[e0:e1].long()
        for o, p in zip(out, per_edge):
            o[i, : e1 - e0] = p[e0:e1]
    return NN, out


def oracle_check(sc, Z, nxt_before, t_local, n_iter, a, b, n_neg, seed, rows, geom=0):
    """Production gradient (in-kernel negatives) of the rows `rows` vs the oracle evaluated on the negatives the kernel
    draws (umap.py:236-292).  Returns the max error relative to max |g|."""
    import oracle.ref_torch as R
    from torchdr_amd import _lib

    grad = sc.grad(Z, t_local, n_iter, a, b, n_neg, neg=None, seed=seed, geom=geom).cpu()
    n = sc.n_rows
    act = sc.records(sc.B)[2][t_local * sc.S].to(torch.int32).cuda()
    nuse = torch.clamp(act * 5, max=n_neg).to(torch.int32).contiguous()
    width = max(int(nuse.max()), 1)
    neg = torch.empty((n, width), dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().tdr_umap_debug_negatives(seed, n_iter, sc.n_total, sc.row0, n, _lib.ptr(nuse), sc.S, width, _lib.ptr(neg),
                                                   _lib.stream_ptr()), "debug_negatives")
    neg = neg.cpu()[rows]
    assert torch.equal((neg >= 0).sum(1), nuse.cpu().long()[rows])
    neg = torch.where(neg >= 0, neg, torch.zeros_like(neg))  # unused slots: masked by the oracle's count
    if width < n_neg:
        neg = torch.cat([neg, torch.zeros((rows.numel(), n_neg - width), dtype=torch.int64)], 1)
    rowptr, cols = sc.rowptr.cpu(), sc.cols.cpu()
    NN, (ep_p, nx_p) = csr_rows_to_padded(rowptr, cols, [sc.eps_per.cpu(), nxt_before.cpu()], rows, [float("inf"), float("inf")])
    ga, gr, act_o = R.umap_gradients(Z.cpu(), NN, ep_p, nx_p, neg, n_iter, a, b, rows=rows + sc.row0)
    assert torch.equal(act_o.sum(1), act.cpu().long()[rows])
    ref = ga + gr
    err = float((grad[rows] - ref).abs().max() / ref.abs().max())
    return err


@pytest.mark.parametrize("S", [1, 2, 4, 8])
@pytest.mark.parametrize("nc", [2, 3, 5, 16])
def test_in_kernel_negatives_vs_oracle_fixture_graph(S, nc):
    """The production path (counter-hash negatives drawn inside the kernel, every slice count) on the reference's own
    affinity graph of the umap_step fixture."""
    g = load("umap_step")
    a, b, T = float(g["a"]), float(g["b"]), int(g["max_iter"])
    rowptr, cols, vals = (t.cuda() for t in padded_to_csr(g["Psym"], g["Isym"]))
    n = g["X"].shape[0]
    eps_per, nxt = prepare(vals, T)
    sc = Sched(rowptr, cols, eps_per, n, 32, S, nc=nc)
    gen = torch.Generator().manual_seed(11)
    Z = (torch.randn(n, nc, generator=gen) * 3).cuda().contiguous()
    rows = torch.arange(n)
    for t0 in (0, 32):
        before = nxt.clone()
        sc.build(nxt, t0, 32)
        for tl in (0, 13, 31) if t0 == 0 else (5,):
            # counters as they stood at iteration t0 + tl: replay the recurrence on the host
            nb = before.cpu().clone()
            ep = eps_per.cpu()
            for t in range(t0, t0 + tl):
                act = nb <= np.float32(t + 1)
                nb[act] += ep[act]
            for geom in (0, 1, 2, 3):
                err = oracle_check(sc, Z, nb, tl, t0 + tl, a, b, 50, 1234567 + S, rows, geom=geom)
                assert err < 1e-5, (S, nc, t0, tl, geom, err)


@pytest.mark.parametrize("S", [2, 4, 8])
@pytest.mark.parametrize("nc", [2, 3, 4, 7, 32])
def test_joint_slice_launch_is_bit_identical(S, nc):
    """geom & 16: all slices in ONE launch (workgroup b takes slice (b % 8) / (8 / S), i.e. one slice per XCD group) plus
    the combine kernel -- the same gradient, bit for bit, as one launch per slice (same partial sums, added in the same
    order), for every lane geometry, with injected and with in-kernel negatives."""
    n = 5000
    rowptr, cols, vals = random_graph(n, seed=31 + S)
    eps_per, nxt = prepare(vals.cuda(), 200)
    sc = Sched(rowptr.cuda(), cols.cuda(), eps_per, n, 32, S, nc=nc)
    sc.build(nxt, 0, 32)
    gen = torch.Generator().manual_seed(5)
    Z = (torch.randn(n, nc, generator=gen) * 3).cuda().contiguous()
    neg = torch.randint(0, n - 1, (n, 40), generator=gen)
    neg = (neg + (neg >= torch.arange(n)[:, None]).long()).cuda()
    for tl in (0, 17):
        for geom in (0, 1):
            for inj in (None, neg):
                a = sc.grad(Z, tl, tl, 1.577, 0.895, 40, neg=inj, seed=99, geom=geom)
                b = sc.grad(Z, tl, tl, 1.577, 0.895, 40, neg=inj, seed=99, geom=geom | 16)
                assert torch.equal(a, b), (S, nc, tl, geom, inj is None)


def test_in_kernel_negatives_vs_oracle_large_and_wide():
    """N = 500k (two L2 slices by the automatic choice) with n_neighbors = 60: n_negatives = 300 > 255 (the per-step
    kernel's 8-bit row header cannot hold it; the scheduled path has no such field).  Oracle on 4096 sampled rows."""
    from torchdr_amd import _lib
    from torchdr_amd.affinity import UMAPAffinity

    n = 500_000
    X = gmm(n, 32, 2.0, seed=7).cuda()
    csr = UMAPAffinity(n_neighbors=60, max_iter=100)(X, return_csr=True)
    del X
    eps_per, nxt = prepare(csr.vals, 500)
    S = int(_lib.lib().tdr_umap_sched_slices(n, 2))
    assert S == 2
    sc = Sched(csr.rowptr, csr.cols, eps_per, n, 32, S)
    for t0 in (0, 32, 64):
        before = nxt.clone()
        sc.build(nxt, t0, 32)
    gen = torch.Generator().manual_seed(1)
    Z = (torch.randn(n, 2, generator=gen) * 4).cuda().contiguous()
    rows = torch.randperm(n, generator=gen)[:4096].sort().values
    err = oracle_check(sc, Z, before, 0, 64, 1.577, 0.895, 300, 99, rows)
    assert err < 1e-5, err
    # the window's other iterations agree with the per-step kernel run on the same counters with the same negatives
    # (injected into both): 17 steps of the recurrence, then compare gradients and counters
    L = _lib.lib()
    nx_step = before.clone()
    ws = torch.empty(8, dtype=torch.int32, device="cuda")
    neg = torch.randint(0, n, (n, 300), generator=torch.Generator(device="cuda").manual_seed(2), device="cuda")
    neg = neg + (neg >= torch.arange(n, device="cuda")[:, None]).long()
    neg.clamp_(max=n - 1)
    for t in range(64, 64 + 18):
        gs = torch.empty((n, 2), device="cuda")
        _lib.check(L.tdr_umap_grad_f32(_lib.ptr(Z), 2, n, 0, n, _lib.ptr(csr.rowptr), _lib.ptr(csr.cols), _lib.ptr(eps_per),
                                       _lib.ptr(nx_step), 1.577, 0.895, t, 5, 300, _lib.ptr(neg), 0, 1.0, 1.0, 1e-3, _lib.ptr(gs), 1,
                                       _lib.ptr(ws), 0, _lib.stream_ptr()), "umap_grad")
        if t in (64, 70, 81):
            gq = sc.grad(Z, t - 64, t, 1.577, 0.895, 300, neg=neg)
            assert torch.allclose(gq, gs, rtol=1e-5, atol=1e-5 * float(gs.abs().max())), t
    del neg


def test_umap_estimator_scheduled_gradients_equal_per_step_kernel():
    """Whole estimator on the scheduled loop; at every one of 70 steps (two window boundaries) the per-step kernel
    (tdr_umap_grad_f32, its own copy of the epoch counters) is evaluated on the same embedding with the same injected
    negatives: gradients agree to 1e-5 and, at the window ends, so do the epoch counters (bit for bit)."""
    import torchdr_amd
    from torchdr_amd import _lib

    n = 4000
    X = gmm(n, 16, 3.0, seed=4).cuda()
    gen = torch.Generator().manual_seed(0)
    seen = {"max_err": 0.0, "steps": 0}

    class Checked(torchdr_amd.UMAP):
        def on_training_step_start(self):
            super().on_training_step_start()
            r = torch.randint(0, n - 1, (n, 50), generator=gen)
            self.neg_indices_ = (r + (r >= torch.arange(n)[:, None]).long()).cuda()

        def _compute_gradients(self):
            t = int(self.n_iter_)
            if t == 0:
                self._nx_step = self.epoch_of_next_sample.clone()
            elif t % 32 == 0:   # a window has just ended: the scheduled counters stand at iteration t
                assert torch.equal(self._nx_step, self.epoch_of_next_sample)
            grad, rows_only = super()._compute_gradients()
            gs = torch.empty_like(grad)
            ws = torch.empty(8, dtype=torch.int32, device="cuda")
            csr = self._csr
            _lib.check(_lib.lib().tdr_umap_grad_f32(
                _lib.ptr(self.embedding_), 2, n, 0, n, _lib.ptr(csr.rowptr), _lib.ptr(self._loop_cols),
                _lib.ptr(self.epochs_per_sample), _lib.ptr(self._nx_step), float(self._a), float(self._b), t, 5, 50,
                _lib.ptr(self.neg_indices_), 0, 1.0, 1.0, 1e-3, _lib.ptr(gs), 1, _lib.ptr(ws), 0, _lib.stream_ptr()), "umap_grad")
            err = float((grad - gs).abs().max() / gs.abs().max())
            seen["max_err"] = max(seen["max_err"], err)
            seen["steps"] += 1
            return grad, rows_only

    Z = Checked(n_neighbors=10, max_iter=70, random_state=0).fit_transform(X)
    assert seen["steps"] == 70 and seen["max_err"] < 1e-5, seen
    assert bool(torch.isfinite(Z).all())


@pytest.mark.parametrize("geom,momentum", [(0, 0.0), (16, 0.0), (16, 0.6)])
@pytest.mark.parametrize("use_graph", [0, 1])
def test_loop_runner_equals_the_call_by_call_sequence(use_graph, geom, momentum):
    """tdr_umap_loop_run (whole windows enqueued at once, replayed as HIP graphs, iteration base in device memory)
    against the same windows issued call by call (build, S gradient passes, tdr_sgd_step_f32): bit-identical embedding
    and epoch counters after 70 iterations (windows 32 + 32 + 6), squared gradient norms at the inspected iterations.
    geom 16 = the joint launch, which the loop object finishes with ONE combine-and-step kernel (round 4), with and without
    momentum."""
    import ctypes

    from torchdr_amd import _lib

    L = _lib.lib()
    n, T, ci = 20000, 70, 25
    gen = torch.Generator().manual_seed(3)
    rowptr, cols, vals = random_graph(n, seed=9, hub=300)
    eps_csr, _ = prepare(vals.cuda(), 200)
    cols_p, eps_p = layout(rowptr.cuda(), cols.cuda(), eps_csr)
    rowptr = rowptr.cuda()
    lr = torch.linspace(1.0, 0.0, T + 1)[:T].contiguous()
    Z0 = (torch.randn(n, 2, generator=gen) * 3).cuda()
    a, b, seed, S = 1.577, 0.895, 4242, 2

    # call by call
    sc = Sched(rowptr, cols_p, eps_p, n, 32, S)
    Z, nxt = Z0.clone(), eps_p.clone()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    mom_a = torch.zeros_like(Z0) if momentum else None
    mom_b = torch.zeros_like(Z0) if momentum else None
    norms, snaps = {}, {}
    for t0 in range(0, T, 32):
        nw = min(32, T - t0)
        sc.build(nxt, t0, nw)
        for tl in range(nw):
            g = sc.grad(Z, tl, t0 + tl, a, b, 150, neg=None, seed=seed, geom=geom)
            if (t0 + tl) % ci == 0:
                norms[(t0 + tl) // ci] = float((g.double() ** 2).sum())
            _lib.check(L.tdr_sgd_step_f32(_lib.ptr(Z), _lib.ptr(g), _lib.ptr(mom_a), Z.numel(), float(lr[t0 + tl]), momentum,
                                          1 if t0 + tl == 0 else 0, _lib.ptr(flag), t0 + tl, _lib.stream_ptr()), "sgd")
            if (t0 + tl) % ci == 0:
                snaps[t0 + tl] = Z.clone()
    # loop object
    sc2 = Sched(rowptr, cols_p, eps_p, n, 32, S)
    Z2, nxt2 = Z0.clone(), eps_p.clone()
    grad = torch.empty((n, 2), device="cuda")
    lr_d, norm2 = lr.cuda(), torch.zeros(T // ci + 2, device="cuda")
    flag2, scratch = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(16, dtype=torch.int32, device="cuda")
    d = _lib.UmapLoopDesc()
    d.Z, d.nc, d.n_total, d.row0, d.n_rows = _lib.ptr(Z2), 2, n, 0, n
    d.rowptr, d.cols, d.eps_per, d.next = _lib.ptr(rowptr), _lib.ptr(cols_p), _lib.ptr(eps_p), _lib.ptr(nxt2)
    d.blk_base, d.list, d.hdr, d.err = _lib.ptr(sc2.blk_base), _lib.ptr(sc2.list), _lib.ptr(sc2.hdr), _lib.ptr(sc2.err)
    d.acc, d.grad, d.mom_buf = _lib.ptr(sc2.acc), _lib.ptr(grad), _lib.ptr(mom_b)
    d.a, d.b, d.neg_rate, d.n_negatives, d.seed = a, b, 5, 150, seed
    d.exag, d.rep, d.eps, d.n_slices, d.block_iters = 1.0, 1.0, 1e-3, S, 32
    d.lr_table, d.max_iter, d.momentum, d.first_iter, d.check_interval = _lib.ptr(lr_d), T, momentum, 0, ci
    snap = torch.zeros((n, 2), device="cuda")
    d.norm2, d.snap, d.nan_flag, d.scratch, d.gather, d.gather_ctx, d.geom = (_lib.ptr(norm2), _lib.ptr(snap), _lib.ptr(flag2), _lib.ptr(scratch),
                                                                               None, None, geom)
    h = ctypes.c_void_p()
    _lib.check(L.tdr_umap_loop_create(ctypes.byref(h), ctypes.byref(d)), "create")
    side = torch.cuda.Stream()       # graphs cannot be captured on the legacy default stream
    side.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(side):
            _lib.check(L.tdr_umap_loop_run(h, 0, 64, use_graph, _lib.stream_ptr()), "run")   # two replays of the 32-window
            _lib.check(L.tdr_umap_loop_run(h, 64, 6, use_graph, _lib.stream_ptr()), "run")
        torch.cuda.synchronize()
        assert torch.equal(Z2, Z) and torch.equal(nxt2, nxt)
        if momentum:
            assert torch.equal(mom_b, mom_a)
        assert torch.equal(snap, snaps[(T - 1) // ci * ci])     # embedding right after the last inspected iteration
        for k, v in norms.items():
            assert abs(float(norm2[k]) - v) <= 1e-5 * v
        # on the default stream a capture request degrades to plain launches instead of failing
        _lib.check(L.tdr_umap_loop_run(h, 0, 1, 1, None), "run on the default stream")
        torch.cuda.synchronize()
    finally:
        L.tdr_umap_loop_destroy(h)
    assert int(sc2.err.item()) == 0 and int(flag2.item()) == 0
    # argument errors
    assert L.tdr_umap_loop_run(None, 0, 1, 0, None) == -1


@pytest.mark.parametrize("momentum", [0.0, 0.6])
def test_fused_combine_and_sgd_step_changes_nothing(momentum):
    """Stock estimator: the joint gradient launch leaves the per-slice planes and tdr_umap_sched_step_f32 combines them and
    steps the rows in one kernel.  A subclass with an (empty) end-of-step hook takes the form with the separate combine
    kernel and tdr_sgd_step_f32.  Same embedding bit for bit, with and without momentum, across two schedule windows."""
    import torchdr_amd
    from torchdr_amd.neighbor_embedding import umap as umod

    n = 400_000          # two L2 slices at nc = 2 (the joint launch needs more than one)
    X = gmm(n, 16, 3.0, seed=6).cuda()

    class Hooked(torchdr_amd.UMAP):
        def on_training_step_end(self):
            super().on_training_step_end()

    from torchdr_amd import config

    kw = dict(n_neighbors=10, max_iter=45, random_state=0, optimizer_kwargs={"momentum": momentum} if momentum else "auto")
    # both in the caller's numbering: the stock estimator would otherwise run its loop in the cluster order of the pruned
    # search (the subclass, whose hook may look at the rows, never does) and draw its negatives by other row keys
    with config.options(RELABEL=False):
        m = torchdr_amd.UMAP(**kw)
        Za = m.fit_transform(X)
        Zb = Hooked(**kw).fit_transform(X)
    assert int(umod._lib.lib().tdr_umap_sched_slices(n, 2)) == 2
    assert torch.equal(Za, Zb) and bool(torch.isfinite(Za).all())


@pytest.mark.parametrize("nc", [1, 5, 10])
def test_umap_with_more_embedding_dimensions(nc):
    """n_components outside {2, 3} (padded kernel instances): the estimator on the scheduled loop, its per-iteration
    gradient against the oracle on the negatives the kernel draws, blobs stay separated in the embedding."""
    import torchdr_amd
    from torchdr_amd.distance import pairwise_distances

    n = 6000
    X = gmm(n, 20, 3.0, seed=12).cuda()
    Z = torchdr_amd.UMAP(n_neighbors=12, n_components=nc, max_iter=120, random_state=0).fit_transform(X)
    assert Z.shape == (n, nc) and bool(torch.isfinite(Z).all())
    labels = (torch.arange(n) % (n // 100)).cuda()
    _, I = pairwise_distances(Z.contiguous(), metric="sqeuclidean", k=5, exclude_diag=True, return_indices=True)
    agree = float((labels[I.long()] == labels[:, None]).float().mean())
    assert agree > (0.5 if nc == 1 else 0.85), agree


# ---- the loop's row numbering (cluster-sorted order of the kNN stage) ---------------------------------------------------
def test_csr_permute_kernel_vs_torch():
    from torchdr_amd import _lib

    n = 3000
    rowptr, cols, vals = random_graph(n, seed=77)
    rowptr, cols, vals = rowptr.cuda(), cols.cuda(), vals.cuda()
    gen = torch.Generator().manual_seed(3)
    perm = torch.randperm(n, generator=gen).cuda()
    inv = torch.empty(n, dtype=torch.int64, device="cuda")
    inv[perm] = torch.arange(n, device="cuda")
    deg = (rowptr[1:] - rowptr[:-1])[perm]
    new_rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    new_rowptr[1:] = deg.cumsum(0)
    nc, nv = torch.empty_like(cols), torch.empty_like(vals)
    p32, i32 = perm.to(torch.int32).contiguous(), inv.to(torch.int32).contiguous()
    _lib.check(_lib.lib().tdr_csr_permute_f32(_lib.ptr(rowptr), _lib.ptr(cols), _lib.ptr(vals), n, _lib.ptr(p32), _lib.ptr(i32),
                                              _lib.ptr(new_rowptr), _lib.ptr(nc), _lib.ptr(nv), _lib.stream_ptr()), "permute")
    erow = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
    src = rowptr[perm][erow] + (torch.arange(erow.numel(), device="cuda") - new_rowptr[erow])
    assert torch.equal(nc.long(), inv[cols[src].long()]) and torch.equal(nv, vals[src])


def test_umap_loop_in_cluster_order_is_the_same_fit():
    """RELABEL: the loop numbers the points in the kNN stage's cluster-sorted order.  (1) It is the fit of the rows handed
    over in that order (same graph, same initial embedding, same negatives -- the sampler is keyed by loop row numbers),
    returned in the caller's order: compared with an unrelabelled fit of X[order].  (2) Same random_state, same embedding,
    bit for bit, on every run (the order itself is canonical)."""
    import torchdr_amd
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.neighbor_embedding import umap as U

    n = 20000
    X = gmm(n, 32, 2.0, seed=3).cuda()
    init = torch.randn(n, 2, generator=torch.Generator().manual_seed(1)).cuda()
    old_mode, old_rel = dbase.PRUNE_MODE, U.RELABEL
    try:
        dbase.PRUNE_MODE = "force"
        U.RELABEL = True
        # three iterations: the clamped forces make the dynamics sign-driven (a row whose force sum is near zero flips by
        # 8 lr on a last-bit difference), so longer runs of the two fits drift apart
        kw = dict(n_neighbors=15, max_iter=3, random_state=0)
        m1 = torchdr_amd.UMAP(init=init, **kw)
        Z1 = m1.fit_transform(X)
        order = m1.loop_order_
        assert order is not None and torch.equal(order.sort().values, torch.arange(n, device="cuda"))
        m1b = torchdr_amd.UMAP(init=init, **kw)
        assert torch.equal(m1b.fit_transform(X), Z1) and torch.equal(m1b.loop_order_, order)
        U.RELABEL = False
        m2 = torchdr_amd.UMAP(init=init[order].contiguous(), **kw)
        Z2 = m2.fit_transform(X[order].contiguous())
        assert m2.loop_order_ is None
    finally:
        dbase.PRUNE_MODE, U.RELABEL = old_mode, old_rel
    # same arithmetic up to the order in which a row's edges are listed (ties of the layout sort) and summed
    err = (Z1[order] - Z2).abs().max(1).values / Z2.abs().max()
    assert float(err.median()) < 1e-5 and float((err > 1e-3).float().mean()) < 0.01, (float(err.median()), float(err.quantile(0.99)), float(err.max()))
    # a subclass that looks at rows during the optimisation keeps the caller's numbering
    class Watching(torchdr_amd.UMAP):
        def on_training_step_end(self):
            super().on_training_step_end()

    try:
        dbase.PRUNE_MODE = "force"
        w = Watching(init=init, **kw)
        w.fit_transform(X)
        assert w.loop_order_ is None
    finally:
        dbase.PRUNE_MODE = old_mode


# ---- round 4: the schedule build on group-ordered loop state ------------------------------------------------------------
def period_class_np(ep):
    b = ep.numpy().view(np.uint32).astype(np.int64)
    k = np.where(b < 0x3F800000, 0, (b - 0x3F800000) >> 21)
    return torch.from_numpy(np.minimum(k, 63))


def group(rowptr, cols, eps_per, n_total, S):
    from torchdr_amd import _lib

    nnz = cols.numel()
    cols_g, eps_g = torch.empty_like(cols), torch.empty_like(eps_per)
    rs_g = torch.empty(nnz, dtype=torch.uint8, device="cuda")
    order_g = torch.empty(nnz, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().tdr_umap_sched_group_f32(_lib.ptr(rowptr), _lib.ptr(cols), _lib.ptr(eps_per), rowptr.numel() - 1, n_total, S,
                                                   _lib.ptr(cols_g), _lib.ptr(eps_g), _lib.ptr(rs_g), _lib.ptr(order_g), _lib.ptr(err),
                                                   _lib.stream_ptr()), "group")
    assert int(err.item()) == 0
    return cols_g, eps_g, rs_g, order_g


class GroupSched(Sched):
    """plan / build on group-ordered state (16-row regions); grad and records are the base class's."""

    def __init__(self, rowptr, cols_rm, eps_rm, n_total, B, S, nc=2, row0=0, stage=0):
        from torchdr_amd import _lib

        self.lib, self.L = _lib, _lib.lib()
        self.rowptr = rowptr
        self.cols, self.eps_per = cols_rm, eps_rm      # row-major (oracle_check reads them)
        self.n_rows, self.n_total, self.B, self.S, self.nc, self.row0, self.stage = rowptr.numel() - 1, n_total, B, S, nc, row0, stage
        self.cols_g, self.eps_g, self.rs_g, self.order_g = group(rowptr, cols_rm, eps_rm, n_total, S)
        nb = (self.n_rows + 15) // 16
        self.nb = nb
        scratch = torch.empty(nb, dtype=torch.int64, device="cuda")
        self.blk_base = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
        _lib.check(self.L.tdr_umap_sched_plan_groups_f32(_lib.ptr(rowptr), _lib.ptr(self.eps_g), self.n_rows, B, _lib.ptr(scratch),
                                                         _lib.ptr(self.blk_base), _lib.stream_ptr()), "plan_groups")
        cap = int(self.blk_base[-1].item())
        self.list = torch.full((cap + 64,), -7, dtype=torch.int32, device="cuda")
        self.hdr = torch.zeros((int(self.L.tdr_umap_sched_hdr_entries(self.n_rows, B, S)), 2), dtype=torch.int32, device="cuda")
        self.err = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.acc = torch.empty((S * self.n_rows, 2 * nc), device="cuda")
        g0 = rowptr[:-1:16]
        e0 = torch.repeat_interleave(g0, torch.cat([g0[1:], rowptr[-1:]]) - g0)
        self.abs_order = e0 + self.order_g.long()       # row-major edge of every group-ordered entry

    def to_group(self, v_rm):
        return v_rm[self.abs_order].contiguous()

    def to_rows(self, v_g):
        _l = self.lib
        out = torch.empty_like(v_g)
        _l.check(self.L.tdr_umap_sched_ungroup_f32(_l.ptr(self.rowptr), _l.ptr(self.order_g), _l.ptr(v_g), self.n_rows, _l.ptr(out),
                                                   _l.stream_ptr()), "ungroup")
        return out

    def build(self, nxt_g, t0, n):
        _l = self.lib
        _l.check(self.L.tdr_umap_sched_build_groups_f32(_l.ptr(self.rowptr), _l.ptr(self.cols_g), _l.ptr(self.eps_g), _l.ptr(self.rs_g),
                                                        _l.ptr(nxt_g), self.n_rows, t0, n, self.S, _l.ptr(self.blk_base), _l.ptr(self.list),
                                                        _l.ptr(self.hdr), _l.ptr(self.err), self.stage, _l.stream_ptr()), "build_groups")
        assert int(self.err.item()) == 0


@pytest.mark.parametrize("S", [1, 2, 8])
def test_group_order_is_a_stable_sort_by_period_class(S):
    n = 3000
    rowptr, cols, vals = random_graph(n, seed=40 + S, hub=1500)
    eps_per, _ = prepare(vals.cuda(), 200)
    cols_p, eps_p = layout(rowptr.cuda(), cols.cuda(), eps_per)
    cols_g, eps_g, rs_g, order_g = (t.cpu() for t in group(rowptr.cuda(), cols_p, eps_p, n, S))
    cp, epp = cols_p.cpu(), eps_p.cpu()
    step = (n - 1 + S - 1) // S
    for g in range((n + 15) // 16):
        e0, e1 = int(rowptr[16 * g]), int(rowptr[min(16 * g + 16, n)])
        o = order_g[e0:e1].long()
        assert torch.equal(o.sort().values, torch.arange(e1 - e0))                   # a permutation of the group's edges
        assert torch.equal(cols_g[e0:e1], cp[e0:e1][o]) and torch.equal(eps_g[e0:e1], epp[e0:e1][o])
        cls = period_class_np(eps_g[e0:e1])
        assert bool((cls[1:] >= cls[:-1]).all())                                      # classes ascending
        same = cls[1:] == cls[:-1]
        assert bool((o[1:][same] > o[:-1][same]).all())                               # stable inside a class
        row_of = torch.searchsorted(rowptr[16 * g + 1: min(16 * g + 16, n) + 1].contiguous(), e0 + o, right=True)
        assert torch.equal((rs_g[e0:e1] & 15).long(), row_of)
        assert torch.equal((rs_g[e0:e1] >> 4).long(), torch.clamp(cols_g[e0:e1].long() // step, max=S - 1))
    # values travel back to the row-major order
    gs = GroupSched(rowptr.cuda(), cols_p, eps_p, n, 32, S)
    v = torch.rand(cols.numel(), device="cuda")
    assert torch.equal(gs.to_rows(gs.to_group(v)), v)


@pytest.mark.parametrize("S", [1, 2, 4, 8])
@pytest.mark.parametrize("B,t0", [(32, 0), (7, 37), (20, 11)])
def test_grouped_schedule_is_the_step_by_step_recurrence(S, B, t0):
    """tdr_umap_sched_build_groups_f32 against the CPU iteration of umap.py:243-247: counters bit-equal, (length, active
    count) of every record, and every (iteration, slice, row) segment equal IN ORDER to the row's firing edges in their
    (period, column) order -- the order a row-sharded fit and a single-process fit must share.  A hub row of 1500 edges
    puts one group beyond the register-resident chunks; a small LDS stage sends many entries down the direct-store path:
    same lists."""
    n = 3000
    rowptr, cols, vals = random_graph(n, seed=3 + S, hub=1500)
    eps_per, _ = prepare(vals.cuda(), 200)
    cols_p, eps_p = layout(rowptr.cuda(), cols.cuda(), eps_per)
    ep_c, cp = eps_p.cpu(), cols_p.cpu()
    nx_c = ep_c.clone()
    for t in range(t0):
        a = nx_c <= np.float32(t + 1)
        nx_c[a] += ep_c[a]
    nx0 = nx_c.clone()
    gs = GroupSched(rowptr.cuda(), cols_p, eps_p, n, B, S)
    nxt_g = gs.to_group(nx0.cuda())
    gs.build(nxt_g, t0, B)
    fires = []
    for t in range(t0, t0 + B):
        a = nx_c <= np.float32(t + 1)
        nx_c[a] += ep_c[a]
        fires.append(a)
    assert torch.equal(gs.to_rows(nxt_g).cpu(), nx_c)
    row_of = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
    step = (n - 1 + S - 1) // S
    sl_of = torch.clamp(cp.long() // step, max=S - 1)
    start, length, act = gs.records(B)
    lst, base = gs.list.cpu().long(), gs.blk_base.cpu()
    for t in range(B):
        want_act = torch.bincount(row_of[fires[t]], minlength=n)
        for s in range(S):
            k = t * S + s
            assert torch.equal(act[k], want_act)
            sel = fires[t] & (sl_of == s)
            assert torch.equal(length[k], torch.bincount(row_of[sel], minlength=n))
            idx = torch.repeat_interleave(start[k], length[k]) + (torch.arange(int(length[k].sum())) -
                                                                   torch.repeat_interleave(length[k].cumsum(0) - length[k], length[k]))
            rows_g = torch.repeat_interleave(torch.arange(n), length[k])
            assert torch.equal(torch.sort(row_of[sel] * n + cp[sel].long()).values, torch.sort(rows_g * n + lst[idx]).values)
            # order inside a segment: by (rank of this firing among the edge's firings of the 4-iteration run, the row's
            # own edge order) -- a key made of the edge alone
            rank = torch.zeros(cp.numel(), dtype=torch.long)
            for tq in range(t - t % 4, t):
                rank += fires[tq].long()
            e_sel = torch.nonzero(sel).flatten()
            key = (row_of[e_sel] * 16 + rank[e_sel]) * cp.numel() + e_sel
            assert torch.equal(lst[idx], cp[e_sel[torch.argsort(key)]].long()), (t, s)
    # segments tile each 16-row region without gaps, in (iteration, slice, row) order
    nb = gs.nb
    pad = nb * 16 - n
    st = torch.cat([start, start[:, -1:].expand(-1, pad) + length[:, -1:].expand(-1, pad)], 1).view(B * S, nb, 16)
    ln = torch.cat([length, torch.zeros((B * S, pad), dtype=torch.long)], 1).view(B * S, nb, 16)
    assert torch.equal(st[:, :, 1:], st[:, :, :-1] + ln[:, :, :-1])
    assert torch.equal(st[0, :, 0], base[:-1])
    assert torch.equal(st[1:, :, 0], st[:-1, :, 15] + ln[:-1, :, 15])
    assert bool((st[-1, :, 15] + ln[-1, :, 15] <= base[1:]).all())
    # the stage size does not enter the result
    used = int((st[-1, -1, 15] + ln[-1, -1, 15]))
    for stage in (512, 4096):
        gs2 = GroupSched(rowptr.cuda(), cols_p, eps_p, n, B, S, stage=stage)
        nxt2 = gs2.to_group(nx0.cuda())
        gs2.build(nxt2, t0, B)
        assert torch.equal(nxt2, nxt_g) and torch.equal(gs2.hdr, gs.hdr) and torch.equal(gs2.list[:used], gs.list[:used])
    # the row-chunk kernel on the same graph: same counters, same records up to the region layout, same segment content
    sc = Sched(rowptr.cuda(), cols_p, eps_p, n, B, S)
    nxt_r = nx0.clone().cuda()
    sc.build(nxt_r, t0, B)
    assert torch.equal(nxt_r.cpu(), nx_c)
    s2, l2, a2 = sc.records(B)
    assert torch.equal(l2, length) and torch.equal(a2, act)


def test_grouped_schedule_does_not_depend_on_the_group_composition():
    """Rows [r0, n) handed over as a graph of their own (what a rank of a row-sharded fit holds): the groups are cut
    elsewhere, every row's segments list the same columns in the same order."""
    n, S, B = 2000, 2, 32
    rowptr, cols, vals = random_graph(n, seed=77, hub=400)
    eps_per, _ = prepare(vals.cuda(), 200)
    cols_p, eps_p = layout(rowptr.cuda(), cols.cuda(), eps_per)
    full = GroupSched(rowptr.cuda(), cols_p, eps_p, n, B, S)
    nx = full.to_group(eps_p.clone())
    full.build(nx, 0, B)
    st_f, ln_f, _ = full.records(B)
    lst_f = full.list.cpu().long()
    for r0 in (5, 23):
        e0 = int(rowptr[r0])
        rp = (rowptr[r0:] - e0).cuda()
        part = GroupSched(rp, cols_p[e0:].contiguous(), eps_p[e0:].contiguous(), n, B, S)
        nxp = part.to_group(eps_p[e0:].clone())
        part.build(nxp, 0, B)
        st_p, ln_p, _ = part.records(B)
        lst_p = part.list.cpu().long()
        assert torch.equal(ln_p, ln_f[:, r0:])
        for k in (0, 1, 17, 40, 63):
            for r in range(0, n - r0, 7):
                a, b = int(st_p[k, r]), int(st_f[k, r0 + r])
                m = int(ln_p[k, r])
                assert torch.equal(lst_p[a:a + m], lst_f[b:b + m]), (r0, k, r)
        assert torch.equal(part.to_rows(nxp), full.to_rows(nx)[e0:])
