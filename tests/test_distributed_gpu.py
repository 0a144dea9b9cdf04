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


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
        # --- row-sharded INPUT (sharded_input=True): each rank hands over ITS rows only; same embedding as the
        #     replicated-input run, bit for bit (the shards are all-gathered first, then the same code runs)
        Zr = torchdr_amd.UMAP(n_neighbors=12, max_iter=20, random_state=0).fit_transform(X)
        Zs = torchdr_amd.UMAP(n_neighbors=12, max_iter=20, random_state=0, sharded_input=True).fit_transform(X[s:e].clone())
        assert Zs.shape == (n, 2) and torch.equal(Zs, Zr)
        ret[rank] = True
    finally:
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
        Zr = torchdr_amd.UMAP(n_neighbors=30, max_iter=60, random_state=0).fit_transform(X)
        assert Zr.shape == (n, 2) and bool(torch.isfinite(Zr).all())
        h = Zr.detach().cpu()
        gathered = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(gathered, h)
        assert all(torch.equal(gathered[0], g) for g in gathered[1:]), "ranks diverged"
        Zs = torchdr_amd.UMAP(n_neighbors=30, max_iter=60, random_state=0, sharded_input=True).fit_transform(X[s:e].clone())
        assert torch.equal(Zs, Zr)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_eight_rank_sharded_umap_on_one_gpu():
    """C4's parallel layout (8 row shards, uneven chunks) end to end: per-rank affinity rows equal the single-process
    graph bit for bit, every rank ends with the same embedding, row-sharded input gives the same result."""
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_c4, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))
