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
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_path_on_one_gpu():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))
