"""Row-sharded path end to end on ONE GPU: two processes share cuda:0 (gloo, host-staged collectives) and
run the real distributed code -- chunked kNN, edge exchange + symmetrisation, per-step row all-gather /
gradient all-reduce.  RCCL itself needs one GPU per rank, so the transport is the only thing not exercised."""

import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_distributed_cpu import _free_port

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret, backend="gloo"):
    # gloo: every rank on device 0 (host-staged collectives); nccl (= RCCL): one device per rank
    local = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(local))
    torch.cuda.set_device(local)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import torchdr_amd
        from tests.conftest import gmm
        from torchdr_amd.affinity import UMAPAffinity
        from torchdr_amd.distributed import chunk_bounds

        n = 3001  # uneven chunks, chunk start not a multiple of 32
        X = gmm(n, 24, 3.0, seed=4).cuda()
        # --- affinity graph: this rank's rows must equal the single-process result
        aff = UMAPAffinity(n_neighbors=12, max_iter=100)
        assert aff.is_multi_gpu and aff.world_size == world
        csr = aff(X, return_csr=True)
        s, e = chunk_bounds(n, rank, world)
        assert csr.n == e - s and csr.row_offset == s
        ref = UMAPAffinity(n_neighbors=12, max_iter=100, distributed=False)(X, return_csr=True)
        b0, b1 = int(ref.rowptr[s]), int(ref.rowptr[e])
        assert torch.equal(csr.rowptr + b0, ref.rowptr[s:e + 1])
        assert torch.equal(csr.cols, ref.cols[b0:b1])
        assert torch.equal(csr.vals, ref.vals[b0:b1])
        # --- the two-stage (screen + rescore) search on a row chunk whose start is not a multiple of 32: separate
        #     query image, shared fp16 scale, self exclusion by global index
        from torchdr_amd.distance import base as dbase

        dbase.SCREEN_MODE = "force"
        try:
            csr_s = UMAPAffinity(n_neighbors=12, max_iter=100)(X, return_csr=True)
            assert dbase.LAST_KNN["path"] == "screen"
        finally:
            dbase.SCREEN_MODE = "auto"
        assert torch.equal(csr_s.rowptr, csr.rowptr) and torch.equal(csr_s.cols, csr.cols)
        assert torch.equal(csr_s.vals, csr.vals)
        # --- row-sharded search WITH cluster-bound pruning: every rank answers one range of the cluster-sorted order and
        #     the rows travel to their owners (all-to-all-v); must equal the single-process exact result bit for bit
        from torchdr_amd.distance import pairwise_distances
        from torchdr_amd.distributed import DistributedContext

        nb = 9001
        Xb = gmm(nb, 32, 3.0, seed=14).cuda()
        dbase.SCREEN_MODE, dbase.PRUNE_MODE = "force", "force"
        try:
            Cs, Is = pairwise_distances(Xb, metric="sqeuclidean", k=10, exclude_diag=True, return_indices=True,
                                        distributed_ctx=DistributedContext())
            assert dbase.LAST_KNN.get("pruned")
            # --- the same search with the ranks KEEPING their range of the cluster-sorted order (what a row-sharded UMAP
            #     asks for): rows and neighbour indices are positions of the order; mapped back through (perm, inv) they
            #     are the exact result.  perm is the same on every rank (each built the index itself).
            info = {"want_loop_order": True}
            Cl, Il = dbase._pairwise(Xb, None, "sqeuclidean", None, True, 10, True, "auto", DistributedContext(), info)
            assert info.get("loop_order") and info["cluster_order"] is not None
            perm_l, inv_l = info["cluster_order"]
            hp = perm_l.cpu()
            gp = [torch.empty_like(hp) for _ in range(world)]
            dist.all_gather(gp, hp)
            assert all(torch.equal(gp[0], g) for g in gp[1:]), "cluster order differs between ranks"
            assert torch.equal(perm_l.long().sort().values, torch.arange(nb, device="cuda"))
            assert torch.equal(inv_l[perm_l.long()].long(), torch.arange(nb, device="cuda"))
        finally:
            dbase.SCREEN_MODE, dbase.PRUNE_MODE = "auto", "auto"
        dbase.SCREEN_MODE = "0"
        try:
            Ce, Ie = pairwise_distances(Xb, metric="sqeuclidean", k=10, exclude_diag=True, return_indices=True)
        finally:
            dbase.SCREEN_MODE = "auto"
        b0, b1 = chunk_bounds(nb, rank, world)
        assert Cs.shape == (b1 - b0, 10)
        assert torch.equal(Is, Ie[b0:b1]) and torch.equal(Cs, Ce[b0:b1])
        mine = perm_l[b0:b1].long()     # caller's rows of this rank's positions
        assert torch.equal(Cl, Ce[mine]) and torch.equal(perm_l[Il.long()], Ie[mine])
        # --- estimators: every rank ends with the same finite embedding
        for cls, kw in ((torchdr_amd.UMAP, dict(n_neighbors=12, max_iter=40)),
                        (torchdr_amd.LargeVis, dict(perplexity=6, max_iter=25)),
                        (torchdr_amd.TSNE, dict(perplexity=6, max_iter=25)),
                        (torchdr_amd.SNE, dict(perplexity=6, max_iter=25)),
                        (torchdr_amd.InfoTSNE, dict(perplexity=6, max_iter=25, n_negatives=30)),
                        (torchdr_amd.COSNE, dict(perplexity=6, max_iter=25, lr=0.05))):
            m = cls(random_state=0, **kw)
            assert m.world_size == world
            Z = m.fit_transform(X)
            assert Z.shape == (n, 2) and torch.isfinite(Z).all()
            h = Z.detach().cpu()
            gathered = [torch.empty_like(h) for _ in range(world)]
            dist.all_gather(gathered, h)
            assert torch.equal(gathered[0], gathered[1]), f"{cls.__name__}: ranks diverged"
            if cls in (torchdr_amd.SNE, torchdr_amd.COSNE):  # no sampling: the sharded run must reproduce the single-process one
                Z1 = cls(random_state=0, distributed=False, **kw).fit_transform(X)
                assert torch.allclose(Z, Z1, rtol=1e-3, atol=1e-4 * float(Z1.abs().max()))
            if cls is torchdr_amd.UMAP:
                # one negative-sampling seed for all ranks, keyed by global row: the sharded fit IS the single-process fit
                Z1 = cls(random_state=0, distributed=False, **kw).fit_transform(X)
                assert torch.equal(Z, Z1), float((Z - Z1).abs().max())
            if cls is torchdr_amd.LargeVis:
                # round 6: the run-permutation sampler serves every world size (each rank pulls the complete gradient of ITS rows
                # and steps them; keyed by global rows; in-edges summed by ascending source): the SAME draws as the single-process
                # fit, so the embeddings agree to the rounding of the affinity's global normaliser (a sum reduced over ranks) --
                # with the independent sampler of rounds 1-5 the two fits only shared a law
                Z1 = cls(random_state=0, distributed=False, **kw).fit_transform(X)
                assert torch.allclose(Z, Z1, rtol=1e-4, atol=1e-5 * float(Z1.abs().max())), float((Z - Z1).abs().max())
        # --- UMAP with the pruned search: ranks keep their range of the cluster-sorted order, the loop runs in that
        #     numbering on every rank (no kNN row exchange, no index broadcast) -- and is still the single-process fit,
        #     bit for bit, in the caller's order
        dbase.SCREEN_MODE, dbase.PRUNE_MODE = "force", "force"
        try:
            mu = torchdr_amd.UMAP(n_neighbors=10, max_iter=40, random_state=0)
            Zp = mu.fit_transform(Xb)
            assert mu.loop_order_ is not None and dbase.LAST_KNN.get("pruned")
            m1 = torchdr_amd.UMAP(n_neighbors=10, max_iter=40, random_state=0, distributed=False)
            Zp1 = m1.fit_transform(Xb)
            assert m1.loop_order_ is not None and torch.equal(m1.loop_order_, mu.loop_order_)
        finally:
            dbase.SCREEN_MODE, dbase.PRUNE_MODE = "auto", "auto"
        assert torch.equal(Zp, Zp1), float((Zp - Zp1).abs().max())
        # --- row-sharded INPUT (sharded_input=True): each rank hands over ITS rows only; same embedding as the
        #     replicated-input run, bit for bit (the shards are all-gathered first, then the same code runs)
        Zr = torchdr_amd.UMAP(n_neighbors=12, max_iter=20, random_state=0).fit_transform(X)
        Zs = torchdr_amd.UMAP(n_neighbors=12, max_iter=20, random_state=0, sharded_input=True).fit_transform(X[s:e].clone())
        assert Zs.shape == (n, 2) and torch.equal(Zs, Zr)
        # --- float64 input, row-sharded (round 4): float64 chunked search, float64 transposed edges, float64 loop.  The graph rows
        #     are the single-process float64 graph's, bit for bit; the UMAP fit (sampler keyed by global row) likewise; the
        #     estimators with a reduction over ranks agree to float64 rounding
        X64 = X.double()
        c64 = UMAPAffinity(n_neighbors=12, max_iter=100)(X64, return_csr=True)
        r64 = UMAPAffinity(n_neighbors=12, max_iter=100, distributed=False)(X64, return_csr=True)
        assert c64.vals.dtype == torch.float64 and r64.vals.dtype == torch.float64
        b0, b1 = int(r64.rowptr[s]), int(r64.rowptr[e])
        assert torch.equal(c64.rowptr + b0, r64.rowptr[s:e + 1]) and torch.equal(c64.cols, r64.cols[b0:b1])
        assert torch.equal(c64.vals, r64.vals[b0:b1]), float((c64.vals - r64.vals[b0:b1]).abs().max())
        assert float((c64.vals - c64.vals.float().double()).abs().max()) > 0      # float64 values, not widened float32 ones
        for cls, kw in ((torchdr_amd.UMAP, dict(n_neighbors=12, max_iter=40)),
                        (torchdr_amd.LargeVis, dict(perplexity=6, max_iter=25)),
                        (torchdr_amd.TSNE, dict(perplexity=6, max_iter=25)),
                        (torchdr_amd.SNE, dict(perplexity=6, max_iter=25)),
                        (torchdr_amd.InfoTSNE, dict(perplexity=6, max_iter=25, n_negatives=30))):
            m64 = cls(random_state=0, **kw)
            Z64 = m64.fit_transform(X64)
            assert Z64.dtype == torch.float64 and torch.isfinite(Z64).all() and m64._dtype == torch.float64, cls.__name__
            h = Z64.detach().cpu()
            gathered = [torch.empty_like(h) for _ in range(world)]
            dist.all_gather(gathered, h)
            assert torch.equal(gathered[0], gathered[1]), f"{cls.__name__} float64: ranks diverged"
            Z1 = cls(random_state=0, distributed=False, **kw).fit_transform(X64)
            if cls is torchdr_amd.UMAP:
                assert torch.equal(Z64, Z1), float((Z64 - Z1).abs().max())
            elif cls in (torchdr_amd.TSNE, torchdr_amd.SNE):
                # no sampling: the single-process fit up to the bandwidth search's own tolerance (1e-6 on the entropy; the
                # row-sharded search starts from a wider bracket -- no global bounds -- and stops at another point inside it)
                assert float((Z64 - Z1).abs().max()) <= 1e-4 * float(Z1.abs().max()), (cls.__name__, float((Z64 - Z1).abs().max()), float(Z1.abs().max()))
            assert float((Z64 - Z64.float().double()).abs().max()) > 0, cls.__name__     # computed in float64
        # --- the peer exchange itself (csrc/tdr_peerx.hip): mapped through HIP IPC (ranks of this test may share the device),
        #     uneven chunks, several column counts, many generations; and the UMAP fits above went through it
        from torchdr_amd.parallel import PeerExchange

        px = PeerExchange.shared(n, 2, torch.device("cuda", local))
        assert px is not None, "peer exchange unavailable (HIP IPC)"
        # UMAP exchanges rows (its step is the rows-only one); an estimator that all-reduces a full gradient or gathers through
        # torch.distributed (the last of the loop above: COSNE) no longer builds an exchange context it would never use
        assert mu.row_exchange_ == "PeerExchange" and m.row_exchange_ == "torch.distributed", (mu.row_exchange_, m.row_exchange_)
        for rnd in range(40):
            nc = 1 + rnd % 3
            Zx = torch.full((n, nc), float("nan"), device="cuda")
            Zx[s:e] = torch.arange(s, e, device="cuda", dtype=torch.float32)[:, None] * (rnd + 1) + torch.arange(nc, device="cuda")
            px.allgather_rows_(Zx)
            want = torch.arange(n, device="cuda", dtype=torch.float32)[:, None] * (rnd + 1) + torch.arange(nc, device="cuda")
            assert torch.equal(Zx, want), rnd
        assert not px.failed()
        ret[rank] = True
    finally:
        from torchdr_amd.parallel import RcclContext

        RcclContext.destroy_shared()    # the process's communicator (one per process, shared by every fit)
        from torchdr_amd.parallel import PeerExchange

        PeerExchange.destroy_shared()
        dist.destroy_process_group()


def test_rccl_context_single_rank():
    """tdr_ctx_*: librccl opened with dlopen, communicator of ONE rank on this GPU (what a 1-GPU box can run): the
    in-place row all-gather and the all-reduce are identities, on a side stream and inside a captured HIP graph of the
    UMAP loop object, whose result must equal the run without a context."""
    import ctypes

    from tests.test_umap_sched_gpu import Sched, layout, prepare, random_graph
    from torchdr_amd import _lib
    from torchdr_amd.parallel import RcclContext

    n = 20000
    ctx = RcclContext.create(n, torch.device("cuda", 0), rank=0, world=1)
    if ctx is None:
        pytest.skip("librccl could not be opened in this environment")
    try:
        Z = torch.randn(n, 2, device="cuda")
        Z0 = Z.clone()
        ctx.allgather_rows_(Z)
        g = torch.arange(10, dtype=torch.float32, device="cuda")
        ctx.allreduce_(g)
        torch.cuda.synchronize()
        assert torch.equal(Z, Z0) and torch.equal(g.cpu(), torch.arange(10, dtype=torch.float32))
        # loop object with the context's all-gather after every step, replayed as a graph
        L = _lib.lib()
        rowptr, cols, vals = random_graph(n, seed=9, hub=300)
        eps_csr, _ = prepare(vals.cuda(), 200)
        cols_p, eps_p = layout(rowptr.cuda(), cols.cuda(), eps_csr)
        rowptr = rowptr.cuda()
        T = 40
        lr = torch.linspace(1.0, 0.0, T + 1)[:T].contiguous().cuda()
        outs = []
        for use_ctx in (False, True):
            sc = Sched(rowptr, cols_p, eps_p, n, 32, 2)
            Zl, nxt = Z0.clone(), eps_p.clone()
            grad, norm2 = torch.empty((n, 2), device="cuda"), torch.zeros(8, device="cuda")
            flag, scratch = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(16, dtype=torch.int32, device="cuda")
            d = _lib.UmapLoopDesc()
            d.Z, d.nc, d.n_total, d.row0, d.n_rows = _lib.ptr(Zl), 2, n, 0, n
            d.rowptr, d.cols, d.eps_per, d.next = _lib.ptr(rowptr), _lib.ptr(cols_p), _lib.ptr(eps_p), _lib.ptr(nxt)
            d.blk_base, d.list, d.hdr, d.err = _lib.ptr(sc.blk_base), _lib.ptr(sc.list), _lib.ptr(sc.hdr), _lib.ptr(sc.err)
            d.acc, d.grad, d.mom_buf = _lib.ptr(sc.acc), _lib.ptr(grad), None
            d.a, d.b, d.neg_rate, d.n_negatives, d.seed = 1.577, 0.895, 5, 150, 77
            d.exag, d.rep, d.eps, d.n_slices, d.block_iters = 1.0, 1.0, 1e-3, 2, 32
            d.lr_table, d.max_iter, d.momentum, d.first_iter, d.check_interval = _lib.ptr(lr), T, 0.0, 0, 50
            d.norm2, d.snap, d.nan_flag, d.scratch = _lib.ptr(norm2), None, _lib.ptr(flag), _lib.ptr(scratch)
            d.gather, d.gather_ctx, d.geom = (ctx.gather_fn, ctx.handle, 0) if use_ctx else (None, None, 0)
            h = ctypes.c_void_p()
            _lib.check(L.tdr_umap_loop_create(ctypes.byref(h), ctypes.byref(d)), "create")
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            try:
                with torch.cuda.stream(side):
                    _lib.check(L.tdr_umap_loop_run(h, 0, T, 1, _lib.stream_ptr()), "run")
                torch.cuda.synchronize()
            finally:
                L.tdr_umap_loop_destroy(h)
            assert int(flag.item()) == 0 and int(sc.err.item()) == 0
            outs.append(Zl)
        assert torch.equal(outs[0], outs[1])
    finally:
        ctx.destroy()


def test_two_rank_sharded_path_on_one_gpu():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
def test_two_rank_sharded_path_over_rccl():
    """The same checks with backend nccl (= RCCL over xGMI), one device per rank: chunked / pruned / loop-order searches,
    the edge all-to-all of the symmetrisation, every estimator (UMAP through the C loop object with `tdr_ctx_allgather_rows`
    enqueued on the compute stream, uneven chunks: grouped broadcasts), row-sharded input -- all against the
    single-process results."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, "nccl"), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


def _worker_rccl_even(rank, world, port, ret):
    """Even chunks over RCCL: the in-place ncclAllGather form of the per-iteration exchange, LargeVis / TSNE gradient
    all-reduce, and the UMAP loop in cluster order at a size where the default dispatch prunes."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        import torchdr_amd
        from tests.conftest import gmm
        from torchdr_amd.distance import base as dbase

        n = 70000 - 70000 % world
        X = gmm(n, 32, 2.0, seed=5).cuda()
        m = torchdr_amd.UMAP(n_neighbors=15, max_iter=60, random_state=0)
        Z = m.fit_transform(X)
        assert m.world_size == world and dbase.LAST_KNN.get("pruned") and m.loop_order_ is not None
        Z1 = torchdr_amd.UMAP(n_neighbors=15, max_iter=60, random_state=0, distributed=False).fit_transform(X)
        assert torch.equal(Z, Z1), float((Z - Z1).abs().max())
        s, e = n * rank // world, n * (rank + 1) // world
        Zs = torchdr_amd.UMAP(n_neighbors=15, max_iter=60, random_state=0, sharded_input=True).fit_transform(X[s:e].clone())
        assert torch.equal(Zs, Z)
        for cls, kw in ((torchdr_amd.LargeVis, dict(perplexity=6, max_iter=25)), (torchdr_amd.TSNE, dict(perplexity=6, max_iter=25))):
            Zc = cls(random_state=0, **kw).fit_transform(X[:6000].contiguous())
            assert bool(torch.isfinite(Zc).all())
            h = Zc.detach().clone()
            g = [torch.empty_like(h) for _ in range(world)]
            dist.all_gather(g, h)
            assert all(torch.equal(g[0], t) for t in g[1:]), f"{cls.__name__}: ranks diverged"
        ret[rank] = True
    finally:
        from torchdr_amd.parallel import RcclContext

        RcclContext.destroy_shared()    # the process's communicator (one per process, shared by every fit)
        from torchdr_amd.parallel import PeerExchange

        PeerExchange.destroy_shared()
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
@pytest.mark.parametrize("world", [2, 4, 8])
def test_even_chunks_over_rccl(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_rccl_even, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no rank in the environment (the driver's invocation) starts two ranks itself --
    over RCCL when the node has two devices, else sharing the device over gloo -- and prints ONE line with n_gpus = 2 and
    the per-phase split."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    # at this size the index has 70 clusters for 700 blobs and the default dispatch would (rightly) not prune: force the
    # pruned, cluster-ordered path the headline size takes
    env.update(TDR_KNN_PRUNE="force", TDR_KNN_SCREEN="force")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--npoints", "70000", "--steps", "1",
                          "--warmup", "1", "--max-iter", "100", "--no-cpu-baseline", "--no-knn-variants"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["scaling"] == "strong"
    assert rec["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert rec["devices_shared"] == (torch.cuda.device_count() < 2)
    assert rec["loop_in_cluster_order"] and len(rec["hbm_peak_gb"]) == 2
    for name in ("shard gather (all-gather of X)", "knn: pruned scan + rescoring", "symmetrise: edge exchange (all-to-all)", "loop"):
        assert name in rec["phases_ms"], sorted(rec["phases_ms"])


def test_first_contact_script_runs_end_to_end(tmp_path):
    """tools/gpu_first_contact.sh -- the one command for the first node with more than one GPU -- at a small size on THIS box (ranks share
    the device when it has one): every step returns 0, the per-N summary holds a measured line per (gpus, exchange) with its phase
    split and the transport that ran, the emulated rank share beside it, and their ratio."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(FC_SKIP_TESTS="1", FC_SIZES="70000:32", FC_WORLDS="1 2", FC_MAX_ITER="60", TDR_KNN_PRUNE="force", TDR_KNN_SCREEN="force")
    out = subprocess.run(["bash", os.path.join(root, "tools", "gpu_first_contact.sh"), str(tmp_path)], env=env, capture_output=True, text=True,
                         timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    steps = [json.loads(ln) for ln in open(tmp_path / "steps.jsonl")]
    assert steps and all(s_["rc"] == 0 for s_ in steps), steps
    rec = json.load(open(tmp_path / "first_contact_n70000.json"))
    got = {(m["gpus"], m["exchange_requested"]) for m in rec["measured"]}
    assert got == {(1, "rccl"), (2, "rccl"), (2, "peer")}, got
    for m in rec["measured"]:
        assert m["ms_per_step"] > 0 and "loop" in m["phases_ms"]
        if m["gpus"] == 2:
            assert m["row_exchange"] in ("PeerExchange", "RcclContext", "torch.distributed") and m["measured_over_emulated"] > 0
    assert rec["emulated"] and rec["emulated"][0]["world"] == 2 and rec["emulated"][0]["exchange_bytes"]["received_per_iteration"] > 0


def _worker_c4(rank, world, port, ret):
    """BASELINE config C4's shape (D = 256, k = 30, rows sharded over 8 ranks) at a size one GPU can host 8 times."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torchdr_amd
        from tests.conftest import gmm
        from torchdr_amd.affinity import UMAPAffinity
        from torchdr_amd.distributed import chunk_bounds

        n = 8003   # 8003 = 8 * 1000 + 3: three ranks own one row more, no chunk starts on a multiple of 32
        X = gmm(n, 256, 2.0, seed=44).cuda()
        s, e = chunk_bounds(n, rank, world)
        csr = UMAPAffinity(n_neighbors=30, max_iter=100)(X, return_csr=True)
        ref = UMAPAffinity(n_neighbors=30, max_iter=100, distributed=False)(X, return_csr=True)
        b0, b1 = int(ref.rowptr[s]), int(ref.rowptr[e])
        assert csr.n == e - s and csr.row_offset == s
        assert torch.equal(csr.rowptr + b0, ref.rowptr[s:e + 1]) and torch.equal(csr.cols, ref.cols[b0:b1])
        assert torch.equal(csr.vals, ref.vals[b0:b1])
        mr = torchdr_amd.UMAP(n_neighbors=30, max_iter=60, random_state=0)
        Zr = mr.fit_transform(X)
        assert Zr.shape == (n, 2) and bool(torch.isfinite(Zr).all())
        # eight ranks' rows travelled as direct peer writes (csrc/tdr_peerx.hip, mapped through HIP IPC): still the single-process fit
        assert mr.row_exchange_ == "PeerExchange", mr.row_exchange_
        Z1 = torchdr_amd.UMAP(n_neighbors=30, max_iter=60, random_state=0, distributed=False).fit_transform(X)
        assert torch.equal(Zr, Z1), float((Zr - Z1).abs().max())
        h = Zr.detach().cpu()
        gathered = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(gathered, h)
        assert all(torch.equal(gathered[0], g) for g in gathered[1:]), "ranks diverged"
        Zs = torchdr_amd.UMAP(n_neighbors=30, max_iter=60, random_state=0, sharded_input=True).fit_transform(X[s:e].clone())
        assert torch.equal(Zs, Zr)
        ret[rank] = True
    finally:
        from torchdr_amd.parallel import RcclContext

        RcclContext.destroy_shared()    # the process's communicator (one per process, shared by every fit)
        from torchdr_amd.parallel import PeerExchange

        PeerExchange.destroy_shared()
        dist.destroy_process_group()


def test_eight_rank_sharded_umap_on_one_gpu():
    """C4's parallel layout (8 row shards, uneven chunks) end to end: per-rank affinity rows equal the single-process
    graph bit for bit, every rank ends with the same embedding, row-sharded input gives the same result."""
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_c4, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))
