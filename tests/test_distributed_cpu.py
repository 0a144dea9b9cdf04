"""N > 1 host logic on CPU: world_size-2 gloo processes exercise the exchange steps of the row-sharded
path (edge routing all-to-all-v, row all-gather, gradient all-reduce) that RCCL runs on the GPUs."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_oracle_golden import load


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torchdr_amd.distributed import DistributedContext, chunk_bounds
        from torchdr_amd.parallel import allgather_rows, allreduce_, exchange_transposed_edges

        a = load("affinity")
        P, I = a["umap10_P"][:n_rows], a["umap10_I"][:n_rows].clamp(max=n_rows - 1)
        ctx = DistributedContext()
        assert ctx.is_initialized and ctx.world_size == world and ctx.rank == rank
        s, e = ctx.compute_chunk_bounds(n_rows)
        assert (s, e) == chunk_bounds(n_rows, rank, world)
        er, ec, ev = exchange_transposed_edges(P[s:e], I[s:e], s, n_rows, world)
        # expected: every edge (i -> j) of OTHER ranks whose target j is owned here
        owner = DistributedContext.get_rank_for_indices(torch.arange(n_rows), n_rows, world)
        ii = torch.arange(n_rows).repeat_interleave(P.shape[1])
        jj = I.reshape(-1).long()
        vv = P.reshape(-1)
        m = (owner[jj] == rank) & (owner[ii] != rank)
        exp = sorted(zip((jj[m] - s).tolist(), ii[m].tolist(), vv[m].tolist()))
        got = sorted(zip(er.tolist(), ec.tolist(), ev.tolist()))
        assert got == exp, f"rank {rank}: routed edges differ"
        # float64 graph (float64 input, round 4): the values travel in float64, not through a float32 buffer
        P64 = P.double() * (1.0 + 2.0 ** -40)
        er64, ec64, ev64 = exchange_transposed_edges(P64[s:e], I[s:e], s, n_rows, world)
        assert ev64.dtype == torch.float64
        assert sorted(zip(er64.tolist(), ec64.tolist(), ev64.tolist())) == sorted(zip((jj[m] - s).tolist(), ii[m].tolist(), P64.reshape(-1)[m].tolist()))
        # all-gather of uneven row chunks reassembles the full matrix
        full = torch.arange(n_rows * 2, dtype=torch.float32).reshape(n_rows, 2)
        out = allgather_rows(full[s:e].clone(), n_rows, world)
        assert torch.equal(out, full)
        # in-place form (per-iteration exchange of updated rows): equal chunks -> send buffer aliases its slot
        from torchdr_amd.parallel import allgather_rows_

        for n_eq in (n_rows - n_rows % world, n_rows):               # equal chunks, then (possibly) uneven ones
            ref = torch.arange(n_eq * 2, dtype=torch.float32).reshape(n_eq, 2)
            s2, e2 = chunk_bounds(n_eq, rank, world)
            mine_only = torch.full_like(ref, -1.0)
            mine_only[s2:e2] = ref[s2:e2]
            assert allgather_rows_(mine_only, s2, e2 - s2, world) is mine_only and torch.equal(mine_only, ref)
        g = torch.full((4, 2), float(rank + 1))
        allreduce_(g)
        assert torch.equal(g, torch.full((4, 2), float(sum(range(1, world + 1)))))
        # kNN rows computed for arbitrary global rows travel to their owners (pruned sharded search): every rank holds
        # the rows of an interleaved subset, afterwards each owns its contiguous chunk in row order
        from torchdr_amd.parallel import allreduce_max_, broadcast_, exchange_rows_to_owners

        k = 5
        allC = torch.arange(n_rows * k, dtype=torch.float32).reshape(n_rows, k)
        allI = (torch.arange(n_rows * k, dtype=torch.int64).reshape(n_rows, k) % 1000).to(torch.int32)
        mine = torch.arange(rank, n_rows, world)                    # rows this rank "searched"
        Cc, Ic = exchange_rows_to_owners(mine.to(torch.int32), allC[mine], allI[mine], n_rows, world, s, e - s)
        assert torch.equal(Cc, allC[s:e]) and torch.equal(Ic, allI[s:e])
        v = torch.tensor([float(rank == 1), float(rank), 10.0 - rank], dtype=torch.float64)
        allreduce_max_(v)
        assert v.tolist() == [1.0, float(world - 1), 10.0]
        b = torch.arange(6, dtype=torch.int32) * (1 if rank == 0 else 0)
        broadcast_(b)
        assert b.tolist() == list(range(6))
        # row-sharded INPUT: every rank passes its shard, gets the full block back (one all-gather of the shards)
        from torchdr_amd.parallel import gather_row_shards

        Xfull = torch.arange(n_rows * 3, dtype=torch.float32).reshape(n_rows, 3)
        assert torch.equal(gather_row_shards(Xfull[s:e].clone()), Xfull)
        if n_rows % world:   # shards that do not follow the chunk rule are refused, on every rank alike
            wrong = Xfull[: n_rows // world] if rank == 0 else Xfull[n_rows // world:]
            with pytest.raises(ValueError, match="chunk rule"):
                gather_row_shards(wrong.clone())
        ret[rank] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [600, 599])
def test_exchange_steps_world_size_2(n_rows):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_rows, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


def test_exchange_steps_world_size_3_uneven():
    """The same exchanges over three ranks with chunks of 200 / 200 / 199 rows (the eight-rank layout of BASELINE C4 has
    such a short last chunk)."""
    world = 3
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, 599, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


def test_route_edges_partition():
    """route_edges is a pure partition of the edge list (no communication)."""
    from torchdr_amd.distributed import DistributedContext
    from torchdr_amd.parallel import route_edges

    a = load("affinity")
    P, I = a["umap30_P"], a["umap30_I"]
    n, k = P.shape
    W = 4
    routed = route_edges(P, I, 0, n, W, rank=1)
    assert routed[1] is None
    total = sum(p[0].numel() for p in routed if p is not None)
    owner = DistributedContext.get_rank_for_indices(I.reshape(-1).long(), n, W)
    assert total == int((owner != 1).sum())
    for r, p in enumerate(routed):
        if p is None:
            continue
        assert (DistributedContext.get_rank_for_indices(p[1].long(), n, W) == r).all()


def test_abi_header_matches_library():
    """Every prototype of include/torchdr_amd.h is exported by the built shared object (no compute)."""
    import ctypes

    from torchdr_amd import _lib

    protos = _lib.parse_header()
    assert len(protos) >= 20
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(so, name), f"{name} declared in the header but not exported"
    # host-only queries work without a GPU
    so.tdr_packed_floats.restype = ctypes.c_int64
    so.tdr_packed_floats.argtypes = [ctypes.c_int64, ctypes.c_int]
    assert so.tdr_packed_floats(64, 128) == 2 * (16 * 256 + 64)
    assert so.tdr_packed_floats(64, 300) == 0
    so.tdr_knn_max_k.argtypes = [ctypes.c_int]
    assert so.tdr_knn_max_k(128) >= 90


def test_abi_argument_checks_run_before_any_device_work():
    """Entry points validate on the host and return TDR_ERR_BAD_ARG (-1) / TDR_ERR_UNSUPPORTED (-2) without touching
    the device: callable here, without a GPU (dummy non-null addresses are never dereferenced)."""
    import ctypes

    from torchdr_amd import _lib

    L = _lib.lib()
    p, null = ctypes.c_void_p(64), ctypes.c_void_p(0)
    assert L.tdr_l1_block_f32(null, 8, 4, p, 8, 4, 8, p, 4, null) == -1
    assert L.tdr_l1_block_f32(p, 4, 4, p, 8, 4, 8, p, 4, null) == -1                 # row stride < d
    assert L.tdr_l1_exact_f32(p, 9000, null, 4, 0, p, 9000, null, 0, 4, 0, 8192, 0, p, 4, null) == -2   # third cascade level
    assert L.tdr_l1_exact_f32(p, 8, null, 4, 0, p, 8, p, 2, 4, 0, 8, 0, p, 4, null) == -1                # ldc < nc
    assert L.tdr_topk_merge_cand_f32(p, p, 8, 4, 8, 1300, p, null) == -2             # k > tdr_topk_max_k() = 1024
    assert L.tdr_topk_max_k() == 1024
    assert L.tdr_topk_merge_f32(p, 8, 4, 8, null, null, 0, 0, 5, 4, 0, p, null) == -1   # sqhyperbolic needs the norms
    assert L.tdr_topk_merge_f32(p, 8, 4, 8, null, null, 0, 0, 5, 7, 0, p, null) == -1   # unknown metric
    assert L.tdr_hyperbolic_from_gram_f32(p, 4, 2, 8, p, p, null) == -1              # ld < nd
    assert L.tdr_cosne_pairs_f64(p, 9, 100, 0, 100, 2.0, p, p, 1 << 30, null) == -2  # n_components > 8
    assert L.tdr_cosne_pairs_f64(p, 2, 100, 50, 100, 2.0, p, p, 1 << 30, null) == -1  # chunk outside the point set
    assert L.tdr_cosne_pairs_f64(p, 2, 100, 0, 100, 2.0, p, p, 8, null) == -1        # workspace too small
    assert L.tdr_radam_poincare_f64(p, p, p, p, null, 10, 1, 0.9, 0.999, 1e-8, 0.1, 1.0, null, 0, null) == -2
    # host-only sizing helpers
    assert 1 <= L.tdr_cosne_splits(1_000_000, 1_000_000) <= 64 and L.tdr_cosne_splits(300, 300) == 2
    assert L.tdr_cosne_workspace_bytes(300, 300, 2) == 2 * 300 * 3 * 8


def test_product_fails_loudly_without_gpu():
    from torchdr_amd import UMAP
    from torchdr_amd.distance import pairwise_distances

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU"):
        pairwise_distances(torch.randn(50, 4), k=3)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU"):
        UMAP(n_neighbors=5).fit_transform(torch.randn(100, 4))
    for metric in ("manhattan", "sqhyperbolic"):
        with pytest.raises(RuntimeError, match="no CPU"):
            pairwise_distances(torch.rand(50, 4) * 0.3, k=3, metric=metric)
    from torchdr_amd import COSNE

    with pytest.raises(RuntimeError, match="no HIP device|no CPU"):
        COSNE(perplexity=5, max_iter=3).fit_transform(torch.randn(100, 4))


def test_eval_argument_errors_match_reference():
    """Validation happens before any device work (eval/neighborhood_preservation.py:96-106, knn_labels.py:107-125)."""
    import numpy as np
    import pytest as _pt
    from torchdr_amd.eval import knn_label_accuracy, neighborhood_preservation

    X = np.zeros((10, 3), dtype=np.float32)
    with _pt.raises(ValueError, match="same number of samples"):
        neighborhood_preservation(X, np.zeros((9, 2), dtype=np.float32), K=3)
    with _pt.raises(ValueError, match="must be less than number of samples"):
        neighborhood_preservation(X, np.zeros((10, 2), dtype=np.float32), K=10)
    with _pt.raises(ValueError, match="at least 1"):
        knn_label_accuracy(X, np.zeros(10), k=0)
    with _pt.raises(ValueError, match="same number of samples"):
        knn_label_accuracy(X, np.zeros(9), k=2)


def test_dataloader_streaming_host_logic():
    """utils/dataloader.py without a GPU: metadata, order, tuple / numpy batches, drop_last, integer cast, errors."""
    import numpy as np
    import pytest
    import torch
    from torch.utils.data import DataLoader, TensorDataset

    from torchdr_amd.distance import pairwise_distances
    from torchdr_amd.utils import dataloader_metadata, materialize_dataloader, to_torch

    X = torch.randn(1000, 32, generator=torch.Generator().manual_seed(0))
    dl = DataLoader(TensorDataset(X, torch.arange(1000)), batch_size=256, shuffle=False)
    assert dataloader_metadata(dl) == (1000, 32, torch.float32, torch.device("cpu"))
    t, backend, device = to_torch(dl, return_backend_device=True)
    assert backend == "dataloader" and device == "cpu" and torch.equal(t, X)
    assert materialize_dataloader(DataLoader(TensorDataset(X), batch_size=300, drop_last=True)).shape == (900, 32)
    assert torch.equal(materialize_dataloader(DataLoader(X.numpy(), batch_size=128)), X)
    Xi = torch.arange(60).reshape(20, 3)
    assert materialize_dataloader(DataLoader(TensorDataset(Xi), batch_size=7)).dtype == torch.float32
    with pytest.raises(ValueError, match="DataLoader is empty"):
        materialize_dataloader(DataLoader(TensorDataset(X[:0]), batch_size=4))
    with pytest.raises(ValueError, match="2-D tensors"):
        materialize_dataloader(DataLoader(TensorDataset(torch.zeros(8)), batch_size=4))
    for kw, msg in [(dict(k=None), "k cannot be None"), (dict(k=5, Y=X), "Y must be None"),
                    (dict(k=5, backend="keops"), "only supports FAISS backend")]:
        with pytest.raises(ValueError, match=msg):
            pairwise_distances(dl, **kw)


def test_faiss_config_and_approximate_search_routing():
    """Host logic of the approximate search: `FaissConfig` mirrors the reference's constructor (distance/faiss.py:174-201),
    and only a full self search with k that the IVF kernel serves is routed to it -- everything else is answered exactly."""
    from torchdr_amd.distance import FaissConfig
    from torchdr_amd.distance.base import _ivf_request

    cfg = FaissConfig(index_type="IVF", nlist=256, nprobe=8, temp_memory=2.0, device=1, some_faiss_option=3)
    assert cfg.approximate and cfg.nlist == 256 and cfg.nprobe == 8 and cfg.faiss_kwargs == {"some_faiss_option": 3}
    assert "IVF" in repr(cfg) and not FaissConfig().approximate
    # as in the reference (tests/test_utils.py:193-207): any string constructs, the SEARCH reports it; an unknown metric
    # under a faiss backend gets the Faiss backend's message
    import torch

    from torchdr_amd.distance import pairwise_distances

    bad = FaissConfig(index_type="HNSW")
    with pytest.raises(ValueError, match="Index type.*not supported"):
        pairwise_distances(torch.randn(50, 10), k=5, backend=bad)
    with pytest.raises(ValueError, match="Only.*euclidean.*sqeuclidean.*angular"):
        pairwise_distances(torch.randn(50, 10), k=5, backend="faiss", metric="cosine")
    with pytest.raises(ValueError, match="The 'cosine' distance is not supported"):
        pairwise_distances(torch.randn(50, 10), k=5, backend=None, metric="cosine")
    assert _ivf_request(cfg, True, 15, 50_000, 64, "sqeuclidean") == (256, 8)
    assert _ivf_request(FaissConfig(index_type="IVFPQ", nlist=100000, nprobe=500000), True, 15, 50_000, 64, "euclidean") == (781, 781)
    for args in ((None, True, 15, 50_000, 64, "sqeuclidean"), ("faiss", True, 15, 50_000, 64, "sqeuclidean"),
                 (FaissConfig(), True, 15, 50_000, 64, "sqeuclidean"),      # Flat = exact
                 (cfg, False, 15, 50_000, 64, "sqeuclidean"),               # cross search
                 (cfg, True, None, 50_000, 64, "sqeuclidean"),              # dense form
                 (cfg, True, 15, 50_000, 64, "angular"), (cfg, True, 15, 50_000, 300, "sqeuclidean"),
                 (cfg, True, 15, 2000, 64, "sqeuclidean")):
        assert _ivf_request(*args) is None, args


def test_umap_renumbers_its_loop_only_when_nobody_watches():
    """Host logic of UMAP's cluster-order numbering: eligible for the stock class on one rank with a float32 graph of all
    rows; any overridden hook, a neighbour-exclusion table, injected negatives, row shards or a float64 graph keep the
    caller's numbering.  Without an order from the kNN stage `_relabel` hands back the graph it was given."""
    import torch

    import torchdr_amd
    from torchdr_amd.utils.sparse import CSRAffinity

    def graph(dtype=torch.float32, n_total=None):
        return CSRAffinity(torch.tensor([0, 1, 2, 3]), torch.tensor([1, 2, 0], dtype=torch.int32), torch.ones(3, dtype=dtype),
                           n_total=n_total)

    def prepared(cls=torchdr_amd.UMAP, **kw):
        m = cls(n_neighbors=2, max_iter=3, **kw)
        m._csr, m.n_samples_in_, m.world_size = graph(), 3, 1
        return m

    assert prepared()._relabel_eligible()
    assert not prepared(discard_NNs=True)._relabel_eligible()
    m = prepared()
    m.neg_indices_ = torch.zeros((3, 10), dtype=torch.int64)
    assert not m._relabel_eligible()
    m = prepared()
    m.world_size = 2
    assert not m._relabel_eligible()
    m = prepared()
    m._csr = graph(torch.float64)
    assert not m._relabel_eligible()
    m = prepared()
    m._csr = graph(n_total=6)     # a row shard of a larger graph
    assert not m._relabel_eligible()

    class Watching(torchdr_amd.UMAP):
        def on_training_step_end(self):
            super().on_training_step_end()

    class OwnInit(torchdr_amd.UMAP):
        def _init_embedding(self, X):
            return super()._init_embedding(X)

    assert not prepared(Watching)._relabel_eligible() and not prepared(OwnInit)._relabel_eligible()
    m = prepared()
    m.affinity_in._row_order = None
    assert m._relabel() is m._csr and m.loop_order_ is None


def test_cluster_index_from_broadcast_tables_needs_no_host_read():
    """Row-sharded pruned search: ranks other than 0 assemble their ClusterIndex from broadcast tables
    (`ClusterIndex.__new__`), so `finish()` -- the deferred host read of a locally built index -- must be a no-op there."""
    from torchdr_amd.distance.base import ClusterIndex

    ci = ClusterIndex.__new__(ClusterIndex)
    assert ci.finish() is ci


def test_top_level_names_of_the_path():
    """`from torchdr import X` keeps working for every in-scope name of the reference's package root (`torchdr/__init__.py`)."""
    import torchdr_amd

    for name in ("Affinity", "LogAffinity", "SparseAffinity", "SparseLogAffinity", "EntropicAffinity", "UMAPAffinity",
                 "SymmetricEntropicAffinity", "SinkhornAffinity", "PACMAPAffinity", "AffinityMatcher", "DRModule",
                 "NeighborEmbedding", "NegativeSamplingNeighborEmbedding", "UMAP", "LargeVis", "TSNE", "TSNEkhorn", "SNE",
                 "InfoTSNE", "PACMAP", "COSNE", "pairwise_distances", "eval", "knn_label_accuracy",
                 "neighborhood_preservation"):
        assert hasattr(torchdr_amd, name), name


def test_distributed_context_faiss_config():
    """`DistributedContext.get_faiss_config` (reference distributed/__init__.py:269-309): the caller's settings on the local GPU."""
    from torchdr_amd.distance import FaissConfig
    from torchdr_amd.distributed import DistributedContext

    ctx = DistributedContext(force_enable=True)
    ctx.rank, ctx.world_size, ctx.local_rank = 3, 8, 3
    assert ctx.get_faiss_config().device == 3 and ctx.get_faiss_config().index_type == "Flat"
    cfg = ctx.get_faiss_config(FaissConfig(index_type="IVF", nlist=64, nprobe=4, temp_memory=2.0, device=0, use_float16=True))
    assert (cfg.device, cfg.index_type, cfg.nlist, cfg.nprobe, cfg.temp_memory) == (3, "IVF", 64, 4, 2.0)
    assert cfg.faiss_kwargs == {"use_float16": True}


def test_training_step_modes_follow_the_reference():
    """`AffinityMatcher._training_step` (reference affinity_matcher.py:354-430): closed-form gradients when the class brings
    its own `_compute_gradients` (tuple contract of this package, or the reference's bare tensor of chunk rows) or sets
    `_use_closed_form_gradients` and fills the attractive / repulsive hooks (neighbor_embedding/base.py:236-254); autograd
    of the loss otherwise.  Host logic only: torch.optim.Adam on CPU tensors stands in for the step kernels."""
    import torch

    from torchdr_amd.neighbor_embedding.base import NeighborEmbedding

    def armed(cls):
        m = object.__new__(cls)
        torch.nn.Module.__init__(m)
        m.embedding_ = torch.ones(6, 2)
        m.optimizer_ = torch.optim.SGD([m.embedding_], lr=1.0)
        m._fused_sgd, m._lr_table, m._lr_pos = False, [0.5, 0.5], 0
        m.early_exaggeration_coeff_, m.repulsion_strength = 2.0, 3.0
        m.n_iter_ = torch.tensor(0)
        m.chunk_size_ = 6
        return m

    class Hooks(NeighborEmbedding):
        _use_closed_form_gradients = True

        def _compute_attractive_gradients(self):
            return torch.full((6, 2), 1.0)

        def _compute_repulsive_gradients(self):
            return torch.full((6, 2), -0.5)

    m = armed(Hooks)
    assert m._training_step() is None and m._lr_pos == 1
    assert torch.allclose(m.embedding_, torch.full((6, 2), 1.0 - 0.5 * (2.0 * 1.0 + 3.0 * -0.5)))

    class Bare(NeighborEmbedding):
        def _compute_gradients(self):
            return torch.full((6, 2), 4.0)

    m = armed(Bare)
    m._training_step()
    assert torch.allclose(m.embedding_, torch.full((6, 2), -1.0))

    class Tup(NeighborEmbedding):
        def _compute_gradients(self):
            return torch.full((6, 2), 2.0), False

    m = armed(Tup)
    m._training_step()
    assert torch.allclose(m.embedding_, torch.zeros(6, 2))

    class Loss(NeighborEmbedding):
        def _compute_attractive_loss(self):
            return (self.embedding_ ** 2).sum()

        def _compute_repulsive_loss(self):
            return self.embedding_.sum() * 0.0

    m = armed(Loss)
    loss = m._training_step()
    assert float(loss.detach()) == 2.0 * 12.0 and torch.allclose(m.embedding_.detach(), torch.full((6, 2), 1.0 - 0.5 * 4.0))

    class Unfilled(NeighborEmbedding):
        _use_closed_form_gradients = True

    with pytest.raises(NotImplementedError, match="_compute_attractive_gradients"):
        armed(Unfilled)._training_step()

    class Wrong(NeighborEmbedding):
        def _compute_gradients(self):
            return torch.zeros(5, 2)

    with pytest.raises(RuntimeError, match="Gradient size mismatch"):
        armed(Wrong)._training_step()


def test_get_compute_device_hook():
    """`_get_compute_device` of estimators and affinities (reference base.py:215-217, affinity/base.py:139-160): an explicit
    device is returned as given; "auto" needs a HIP device in this build and says so on a host without one."""
    import torch

    import torchdr_amd
    from torchdr_amd.affinity import EntropicAffinity

    X = torch.zeros(3, 2)
    assert str(torchdr_amd.UMAP(device="cuda:0")._get_compute_device(X)) == "cuda:0"
    assert str(EntropicAffinity(device="cuda:0")._get_compute_device(X)) == "cuda:0"
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):
            torchdr_amd.UMAP()._get_compute_device(X)


def test_utils_names_the_path_imports():
    """The names the reference's in-scope modules import from `torchdr.utils` resolve here too (host helpers)."""
    import torch

    from torchdr_amd.utils import (DistributedContext, compile_if_requested, cross_entropy_loss, matrix_transpose,  # noqa: F401
                                   square_loss, sum_matrix_vector, sum_red, symmetrize_sparse)

    M = torch.arange(6.0).reshape(2, 3)
    assert torch.equal(sum_matrix_vector(M, torch.tensor([10.0, 20.0])), M + torch.tensor([[10.0], [20.0]]))
    assert torch.equal(sum_matrix_vector(M, torch.tensor([1.0, 2.0, 3.0]), transpose=True), M + torch.tensor([[1.0, 2.0, 3.0]]))
    assert matrix_transpose(M).shape == (3, 2) and float(square_loss(M, M + 1)) == 6.0

    def f(x):
        return x + 1

    assert compile_if_requested(f) is f


def test_backend_named_distance_entry_points():
    """`pairwise_distances_torch` / `_faiss` / `_faiss_from_dataloader` (reference distance/torch.py:21, faiss.py:225,477):
    the reference's argument order and errors in front of the one HIP search."""
    import torch

    from torchdr_amd.distance import (LIST_METRICS_FAISS, LIST_METRICS_TORCH, pairwise_distances_faiss,
                                      pairwise_distances_faiss_from_dataloader, pairwise_distances_torch)

    assert "manhattan" in LIST_METRICS_TORCH and LIST_METRICS_FAISS == ["euclidean", "sqeuclidean", "angular"]
    X = torch.zeros(4, 2)
    with pytest.raises(ValueError, match="metrics are supported for FAISS"):
        pairwise_distances_faiss(X, 2, metric="manhattan")
    with pytest.raises(ValueError, match="metrics are supported for FAISS"):
        pairwise_distances_faiss_from_dataloader(torch.utils.data.DataLoader(X, batch_size=2), 2, metric="manhattan")
    with pytest.raises(ValueError, match="not supported"):
        pairwise_distances_torch(X, metric="nope")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="HIP device"):   # no CPU compute path behind any of the names
            pairwise_distances_torch(X, k=2)


def test_public_signatures_match_the_reference():
    """Every in-scope constructor / function takes the reference's parameters, in the reference's order, with the
    reference's defaults (tests/golden/signatures.json, written by make_golden.py from the real package)."""
    import inspect
    import json
    import os

    import torchdr_amd
    import torchdr_amd.affinity as AA
    import torchdr_amd.distance as AD
    import torchdr_amd.eval as AE

    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "signatures.json")))

    def sig(obj):
        ps = inspect.signature(obj).parameters
        return [[k, None if v.default is inspect._empty else repr(v.default)] for k, v in ps.items() if k not in ("self", "kwargs")]

    for name, want in ref.items():
        if name.startswith("affinity."):
            obj = getattr(AA, name.split(".", 1)[1]).__init__
        elif name == "distance.FaissConfig":
            obj = AD.FaissConfig.__init__
        elif name.startswith("distance."):
            obj = getattr(AD, name.split(".", 1)[1])
        elif name.startswith("eval."):
            obj = getattr(AE, name.split(".", 1)[1])
        else:
            obj = getattr(torchdr_amd, name).__init__
        got = sig(obj)
        assert [k for k, _ in got] == [k for k, _ in want], (name, got, want)
        for (k, dg), (_, dw) in zip(got, want):
            assert dg == dw, (name, k, dg, dw)


def test_host_root_searches():
    """`binary_search` / `false_position` (utils/root_search.py:17-143) for arbitrary callables: both bracket by halving /
    doubling and stop at |f| < 1e-6."""
    import torch

    import torchdr_amd
    from torchdr_amd.utils import binary_search, false_position

    t = torch.linspace(0.5, 30, 64, dtype=torch.float64)
    for search in (binary_search, false_position):
        r = search(lambda x: x ** 3 - t, 64, dtype=torch.float64)
        assert float((r ** 3 - t).abs().max()) < 1e-6
    assert torchdr_amd.false_position is false_position and torchdr_amd.binary_search is binary_search


def test_threshold_scan_pass_plan_on_the_host():
    """tdr_knn_screen_flat_plan (host arithmetic of csrc/tdr_knn_screen.hip's flat_plan; no device): the passes of the unpruned
    threshold scan cover every tile position once, grow by at most 4 and by no more than k allows -- a pass that takes a query from
    n seen rows to r n appends ~ (r - 1) (k + (L - k) / r) entries (+ 4 sigma, variance = mean x r for the k-part) to a 256-entry
    region; round 5's first plan grew 7.6x at N = 500k and ignored k (profiles/r05_knn_flat_matrix.jsonl) --, and the visiting
    order's stride is coprime to the tile count.  Shapes the scan does not serve return 0."""
    import ctypes
    import math

    from torchdr_amd import _lib

    L = _lib.lib()
    bounds = (ctypes.c_int32 * 40)()
    stride = ctypes.c_int32(0)

    def plan(n, d, k, terms, LL=128, nq=None):
        nb = L.tdr_knn_screen_flat_plan(nq or n, n, d, k, terms, LL, bounds, 40, ctypes.byref(stride))
        return nb, [int(bounds[i]) for i in range(max(nb, 0))], int(stride.value)

    for n in (131_072, 200_000, 300_000, 500_000, 999_983, 1_000_000, 2_000_000, 16_000_000, 100_000_000):
        n_tiles = (n + 31) // 32
        for k in (1, 5, 15, 30, 48, 64, 100, 120):
            for d, terms in ((16, 1), (64, 3), (128, 1), (128, 2), (128, 3), (200, 1)):
                for LL in (64, 128):
                    if LL < k + 8:
                        continue
                    nb, b, st = plan(n, d, k, terms, LL)
                    assert nb >= 2, (n, d, k, terms, LL, nb)
                    assert b[0] == 8 and b[-1] == n_tiles and all(x < y for x, y in zip(b, b[1:])), b
                    assert all(x % 2 == 0 for x in b[:-1]), b
                    assert 0 < st < n_tiles and math.gcd(st, n_tiles) == 1
                    for lo, hi in zip(b, b[1:]):
                        r = hi / lo
                        assert r <= 4.3, (n, k, b)
                        band = (r - 1.0) * (LL - k) / r
                        mean = (r - 1.0) * k + band
                        # bounds are rounded to even positions: a little slack on the region's 256 entries
                        assert mean + 4.0 * math.sqrt((r - 1.0) * r * k + band) <= 256 * 1.12, (n, k, LL, lo, hi, mean)
    # N = 1M, k = 30: six passes of about x4 (the measured configuration of DESIGN section 3)
    nb, b, _ = plan(1_000_000, 128, 30, 1)
    assert nb == 7 and b == [8, 32, 126, 500, 1984, 7876, 31250], b
    # not served: small databases, D > 256, more than one term above D = 128, lists shorter than k, terms outside 1 .. 3
    assert plan(100_000, 128, 30, 1)[0] == 0
    assert plan(1_000_000, 300, 30, 1)[0] == 0
    assert plan(1_000_000, 200, 30, 3)[0] == 0 and plan(1_000_000, 200, 30, 2)[0] == 0
    assert plan(1_000_000, 128, 30, 1, LL=20)[0] == 0
    assert plan(1_000_000, 128, 30, 4)[0] == 0 and plan(1_000_000, 128, 30, 0)[0] == 0
    assert L.tdr_knn_screen_flat_workspace_bytes(1_000_000, 1_000_000, 128, 30, 2, 128) > 0
    assert L.tdr_knn_screen_flat_workspace_bytes(1_000_000, 1_000_000, 256, 30, 1, 128) > 0


def test_emulated_rank_receives_the_edges_a_real_exchange_would_deliver():
    """utils/emulation.py (one rank of a W-rank fit run alone): the edges `EmulatedRank.prepare(r)` hands rank r are
    exactly the transposes a real all-to-all-v would deliver -- every directed edge (i -> j, v) of another rank's rows whose target j
    lies in r's chunk, as (j local, i global, v), ordered by source rank -- checked against a direct evaluation on random graphs
    with uneven chunks (reference utils/sparse.py:259-309)."""
    from torchdr_amd.distributed import DistributedContext, chunk_bounds
    from torchdr_amd.utils.emulation import EmulatedRank

    n, k, W = 103, 6, 4
    gen = torch.Generator().manual_seed(0)
    idx = torch.randint(0, n, (n, k), generator=gen, dtype=torch.int32)
    val = torch.rand(n, k, generator=gen)
    em = EmulatedRank(W)
    for r in range(W):
        c0, c1 = chunk_bounds(n, r, W)
        em.graphs[r] = (val[c0:c1].clone(), idx[c0:c1].clone(), c0)
    owner = DistributedContext.get_rank_for_indices(torch.arange(n), n, W)
    for r in range(W):
        src, dst, v = em.prepare(r, n)
        c0, c1 = chunk_bounds(n, r, W)
        want = [(int(i), int(j), float(val[i, s])) for i in range(n) if int(owner[i]) != r for s, j in enumerate(idx[i].tolist()) if c0 <= j < c1]
        got = list(zip(src.tolist(), dst.tolist(), [float(x) for x in v]))
        assert sorted(got) == sorted(want)
        assert [int(owner[i]) for i, _, _ in got] == sorted(int(owner[i]) for i, _, _ in got)      # delivered in source-rank order
        assert em.edge_exchange_bytes == 12 * len(want)
