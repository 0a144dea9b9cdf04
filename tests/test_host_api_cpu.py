"""Host-side API behaviour the reference's own unit tests pin down, re-run here without a GPU
(``torchdr/tests/test_utils.py:1188-1246`` validate_tensor / to_torch, ``test_affinity_matcher.py:14-137, 340-395``
constructor and configuration errors)."""

import numpy as np
import pytest
import torch


def test_validate_tensor_and_to_torch():
    from torchdr_amd.utils import to_torch, validate_tensor

    with pytest.raises(ValueError, match="validate_tensor expects a torch.Tensor"):
        validate_tensor(np.random.randn(10, 5))
    X = torch.randn(10, 5)
    assert validate_tensor(X) is X
    assert validate_tensor(torch.randn(10), ensure_2d=True).shape == (10, 1)
    with pytest.raises(ValueError):
        validate_tensor(torch.randn(1, 1), ensure_min_samples=2)
    with pytest.raises(ValueError):
        validate_tensor(torch.randn(1, 1), ensure_min_features=2)
    Xs = torch.randn(10, 5).to_sparse()
    with pytest.raises(ValueError):
        validate_tensor(Xs, accept_sparse=False)
    assert validate_tensor(Xs, accept_sparse=True).is_sparse
    with pytest.raises(ValueError, match="complex tensors are not supported"):
        validate_tensor(torch.randn(10, 5, dtype=torch.cfloat))
    with pytest.raises(ValueError, match="infinite values"):
        validate_tensor(torch.tensor([1.0, float("inf")]))
    Xn = np.random.randn(10, 5)
    Xt, backend, device = to_torch(Xn, return_backend_device=True)
    assert isinstance(Xt, torch.Tensor) and backend == "numpy" and device == "cpu" and torch.equal(Xt, torch.from_numpy(Xn))
    Xt, backend, device = to_torch(X, return_backend_device=True)
    assert backend == "torch" and device == X.device


def test_affinity_matcher_argument_errors():
    from torchdr_amd import AffinityMatcher
    from torchdr_amd.affinity import EntropicAffinity

    aff = EntropicAffinity()
    with pytest.raises(ValueError):
        AffinityMatcher(affinity_in=aff, affinity_out=aff, loss_fn="invalid_loss")
    with pytest.raises(ValueError):
        AffinityMatcher(affinity_in=aff, affinity_out="invalid_affinity")
    with pytest.raises(ValueError):
        AffinityMatcher(affinity_in=None, affinity_out=aff)
    with pytest.raises(ValueError, match="affinity_out must be an Affinity instance"):
        AffinityMatcher(affinity_in=aff, affinity_out=42)
    m = AffinityMatcher(affinity_in=aff, affinity_out=aff, scheduler="invalid_scheduler")
    m.optimizer_, m.lr_ = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=1.0), 1.0
    with pytest.raises(ValueError):
        m._configure_scheduler()
    for bad in ("InvalidOptimizer", dict):
        m = AffinityMatcher(affinity_in=aff, affinity_out=aff, optimizer=bad)
        m.embedding_, m.lr_ = torch.zeros(3, 2), 1.0
        with pytest.raises(ValueError):
            m._configure_optimizer()
    m = AffinityMatcher(affinity_in=aff, affinity_out=aff, init="invalid_init")
    m.device_ = "cpu"
    with pytest.raises(ValueError):
        m._init_embedding(torch.rand(5, 2))
    with pytest.raises(ValueError, match="affinity_out is not set"):
        AffinityMatcher(affinity_in=aff, affinity_out=None)._compute_loss()


def test_matcher_configuration_steps_one_by_one():
    """The reference's unit tests drive the private configuration steps in isolation on host tensors
    (test_affinity_matcher.py:89-246): every init kind, `_set_params`, optimizers and schedulers by name / class / kwargs
    leave real torch objects in `optimizer_` / `scheduler_`."""
    from torch.optim import SGD
    from torch.optim.lr_scheduler import ExponentialLR, StepLR

    from torchdr_amd import AffinityMatcher
    from torchdr_amd.affinity import EntropicAffinity

    aff = EntropicAffinity()
    X = torch.rand(5, 2)
    for init in ("normal", "random", "pca", torch.ones(5, 2), np.ones((5, 2))):
        m = AffinityMatcher(affinity_in=aff, affinity_out=aff, init=init)
        m._init_embedding(X)
        assert m.embedding_.shape == (5, 2)
    m = AffinityMatcher(affinity_in=aff, affinity_out=aff, lr="auto", verbose=True)
    m._set_learning_rate()
    assert m.lr_ == 1.0
    with pytest.raises(ValueError):
        AffinityMatcher(affinity_in=aff, affinity_out=aff)._configure_scheduler()

    def configured(**kw):
        m = AffinityMatcher(affinity_in=aff, affinity_out=aff, **kw)
        m._init_embedding(torch.rand(5, 2))
        assert m._set_params()[0]["params"] is m.embedding_
        m._set_learning_rate()
        m._configure_optimizer()
        return m

    assert isinstance(configured(optimizer="Adam").optimizer_, torch.optim.Adam)
    assert isinstance(configured(optimizer=SGD).optimizer_, SGD)
    m = configured(optimizer="SGD", optimizer_kwargs={"momentum": 0.9})
    assert isinstance(m.optimizer_, SGD) and m.optimizer_.param_groups[0]["momentum"] == 0.9 and m._fused_sgd
    m = configured(scheduler="StepLR", scheduler_kwargs={"step_size": 10})
    m._configure_scheduler()
    assert isinstance(m.scheduler_, StepLR)
    m = configured(scheduler=ExponentialLR, scheduler_kwargs={"gamma": 0.9}, max_iter=4)
    m._configure_scheduler()
    assert isinstance(m.scheduler_, ExponentialLR)
    assert m._lr_table == pytest.approx([1.0, 0.9, 0.81, 0.729])     # the table the loop reads
    m = configured()
    m._configure_scheduler()
    assert m.scheduler_ is None


def test_dataloader_order_and_ivfpq_configuration_errors():
    """Reference tests/test_dataloader.py:274-281 (a shuffling loader is refused: neighbour indices refer to iteration order)
    and :381-389 (IVFPQ with M that does not divide the feature dimension)."""
    from torch.utils.data import DataLoader, TensorDataset

    from torchdr_amd.distance import FaissConfig, pairwise_distances
    from torchdr_amd.utils import materialize_dataloader

    X = torch.randn(300, 8)
    with pytest.raises(ValueError, match="shuffle=False"):
        materialize_dataloader(DataLoader(TensorDataset(X), batch_size=100, shuffle=True))
    with pytest.raises(ValueError, match="shuffle=False"):
        pairwise_distances(DataLoader(TensorDataset(X), batch_size=100, shuffle=True), k=10, return_indices=True)
    from torchdr_amd.distance.faiss import get_dataloader_metadata

    dl = DataLoader(TensorDataset(X), batch_size=64, shuffle=False)
    assert get_dataloader_metadata(dl) is None
    assert torch.equal(materialize_dataloader(dl, device="cpu"), X)
    meta = get_dataloader_metadata(dl)      # reference tests/test_dataloader.py:243-330
    assert (meta["n_samples"], meta["n_features"], meta["dtype"]) == (300, 8, torch.float32)
    with pytest.raises(ValueError, match="must be divisible by M"):
        pairwise_distances(torch.randn(500, 33), k=10, backend=FaissConfig(index_type="IVFPQ", nlist=50, nprobe=10, M=8, nbits=8),
                           return_indices=True)


def test_embedding_padding_and_new_switches_cpu():
    """Host helpers of round 3 that need no device: the zero-padding of embeddings to the next kernel width, and the scoped
    switch of the per-tile bounds."""
    import torch

    from torchdr_amd import config
    from torchdr_amd.affinity.entropic import pad_embedding
    from torchdr_amd.distance import base as dbase

    for nc, w in ((1, 2), (2, 2), (3, 3), (4, 4), (5, 8), (9, 16), (17, 32), (32, 32)):
        Z = torch.randn(7, nc, dtype=torch.float64)
        P = pad_embedding(Z)
        assert P.shape == (7, w) and P.dtype == torch.float32 and P.is_contiguous()
        assert torch.equal(P[:, :nc], Z.float()) and float(P[:, nc:].abs().sum()) == 0.0
    with pytest.raises(NotImplementedError):
        pad_embedding(torch.zeros(3, 33))
    assert dbase._opt("TILE_BOUNDS") is True
    with config.options(TILE_BOUNDS="force"):
        assert dbase._opt("TILE_BOUNDS") == "force"
        with config.options(TILE_BOUNDS=False):
            assert dbase._opt("TILE_BOUNDS") is False
        assert dbase._opt("TILE_BOUNDS") == "force"
    assert dbase._opt("TILE_BOUNDS") is True


def test_round4_switches_and_float64_eligibility_cpu():
    """Host logic of round 4 that needs no device: the new behaviour switches are registered with the scoped override and have
    the documented defaults; which float64 inputs the dense float64 SNEkhorn form takes; the chunk rule shared by the peer
    exchange and the reference (distributed/__init__.py:209-219)."""
    import torch

    from torchdr_amd import TSNEkhorn, config
    from torchdr_amd.affinity.entropic import DensePoints64
    from torchdr_amd.distributed import chunk_bounds
    from torchdr_amd.neighbor_embedding import base as nbase
    from torchdr_amd.neighbor_embedding import umap as umod

    assert umod._opt("GROUPED") is True
    assert umod._opt("SCHED_GEOM") == 16
    assert nbase._opt("PEER_EXCHANGE") == "auto"
    with config.options(GROUPED=False, PEER_EXCHANGE=False, FLAT_SCAN=False):
        assert umod._opt("GROUPED") is False
        assert nbase._opt("PEER_EXCHANGE") is False
    assert umod._opt("GROUPED") is True and nbase._opt("PEER_EXCHANGE") == "auto"
    # round 5: the variants measured slower in round 4 are gone from the library and from the switch table
    for gone in ("BUILD_AHEAD", "FUSE_STEP"):
        assert gone not in config.SWITCHES and not hasattr(umod, gone)
    from torchdr_amd.distance import base as dbase

    assert dbase._opt("FLAT_SCAN") is True
    with pytest.raises(Exception):
        with config.options(NO_SUCH_SWITCH=1):
            pass

    limit = 16384
    assert DensePoints64.eligible(torch.zeros(100, 16, dtype=torch.float64))
    assert DensePoints64.eligible(torch.zeros(limit, 2, dtype=torch.float64))
    assert not DensePoints64.eligible(torch.zeros(limit + 1, 2, dtype=torch.float64))        # dense float64 matrix beyond 2 GiB
    assert not DensePoints64.eligible(torch.zeros(100, 257, dtype=torch.float64))            # no float64 distance kernel
    assert not DensePoints64.eligible(torch.zeros(100, 16, dtype=torch.float32))
    X = torch.zeros(100, 16, dtype=torch.float64)
    assert TSNEkhorn(n_components=2)._float64_ok(X) and TSNEkhorn(n_components=4)._float64_ok(X)
    assert not TSNEkhorn(n_components=5)._float64_ok(X)                                    # register instances: 2, 3, 4
    assert not TSNEkhorn(n_components=2)._float64_ok(torch.zeros(limit + 1, 4, dtype=torch.float64))

    for n, w in ((3001, 2), (8003, 8), (10, 16), (1_000_000, 8)):
        bounds = [chunk_bounds(n, r, w) for r in range(w)]
        assert bounds[0][0] == 0 and bounds[-1][1] == n
        assert all(bounds[r][1] == bounds[r + 1][0] for r in range(w - 1))
        sizes = [e - s for s, e in bounds]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)      # the first n % w ranks hold one more row


def test_threshold_scan_tier_choice_on_the_host(monkeypatch):
    """`distance/base.py:_flat_terms` -- which tier (1 = h.h', 2 = h.h' + h.l', 3 = all three products) and list length the
    unpruned threshold scan takes -- with the pilot launches replaced by fixed answers (no device): tier 0 passed -> one term with
    short lists and NO further pilot; else the one-term, the two-term (counted on the three-term pilot's values) and the three-term
    predictions in this order; D > 128 is served with one term only; small query counts, large k and the switches turn it off."""
    from types import SimpleNamespace

    from torchdr_amd import config
    from torchdr_amd.distance import base as dbase

    calls = []

    def pilot_answers(share_by_key):
        def fake(Q, Y, ops, q0, k, metric, exclude_self, q_offset, tier, pred_L, pred_terms=0):
            calls.append((tier, pred_L, pred_terms))
            return share_by_key[(tier, pred_terms)]
        return fake

    Y = SimpleNamespace(n=1_000_000, d=128)
    args = (None, Y, None, 0, 1_000_000, 30, "sqeuclidean", True, 0)
    ok, bad = 0.01, 0.5
    monkeypatch.setattr(dbase, "_flat_pilot", pilot_answers({(0, 0): bad, (1, 2): bad, (1, 0): bad}))
    assert dbase._flat_terms(*args, 0) == (1, dbase._FLAT_L_SHORT) and calls == []          # tier 0 passed its own pilot
    assert dbase._flat_terms(*args[:5], 60, *args[6:], 0) == (1, dbase._FLAT_L)              # k + 16 beyond the short lists
    assert dbase._flat_terms(*args, 1) == (3, dbase._FLAT_L)                                 # tier 1 passed: three terms without a pilot of their own
    assert calls == [(0, 128, 0), (1, 128, 2)]
    calls.clear()
    assert dbase._flat_terms(*args, -1) == (0, 0) and calls == [(0, 128, 0), (1, 128, 2), (1, 128, 0)]
    monkeypatch.setattr(dbase, "_flat_pilot", pilot_answers({(0, 0): ok, (1, 2): ok, (1, 0): ok}))
    assert dbase._flat_terms(*args, 1) == (1, dbase._FLAT_L)
    monkeypatch.setattr(dbase, "_flat_pilot", pilot_answers({(0, 0): bad, (1, 2): ok, (1, 0): ok}))
    assert dbase._flat_terms(*args, 1) == (2, dbase._FLAT_L) and dbase._flat_terms(*args, -1) == (2, dbase._FLAT_L)
    with config.options(FLAT_TWO_TERMS=False):
        assert dbase._flat_terms(*args, 2) == (3, dbase._FLAT_L) and dbase._flat_terms(*args, -1) == (3, dbase._FLAT_L)
    with config.options(FLAT_FORCE_TERMS=2):
        assert dbase._flat_terms(*args, 0) == (2, dbase._FLAT_L)
    with config.options(FLAT_SCAN=False):
        assert dbase._flat_terms(*args, 0) == (0, 0)
    # 128 < D <= 256: one term or nothing
    Y256 = SimpleNamespace(n=1_000_000, d=256)
    a256 = (None, Y256) + args[2:]
    assert dbase._flat_terms(*a256, 0) == (1, dbase._FLAT_L_SHORT)
    assert dbase._flat_terms(*a256, 1) == (0, 0)                      # one-term pilot fails: two / three terms are not instantiated there
    with config.options(FLAT_FORCE_TERMS=3):
        assert dbase._flat_terms(*a256, 0) == (0, 0)
    # not served: few queries (no pilot slice), k beyond the lists, small databases, D > 256
    assert dbase._flat_terms(*args[:4], 1000, *args[5:], 0) == (0, 0)
    assert dbase._flat_terms(*args[:5], 121, *args[6:], 0) == (0, 0)
    assert dbase._flat_terms(None, SimpleNamespace(n=100_000, d=128), *args[2:], 0) == (0, 0)
    assert dbase._flat_terms(None, SimpleNamespace(n=1_000_000, d=300), *args[2:], 0) == (0, 0)


def test_tile_table_is_skipped_only_where_no_tile_bound_can_help():
    """`ClusterIndex.tiles_hopeless` on hand-made index tables (no device): one Gaussian cut into balls (centres a few units
    apart, radii and neighbour distances several times that) -> the per-tile table is not built; blobs whose balls overlap but
    whose tiles can still be told apart (centre distance 16, radius 6.6, k-th neighbour at 8: sqrt(16^2 + 6.6^2) - 6.6 = 10.7 > 8)
    -> it is; and a few outlier clusters do not pay for a table when nearly every tile would still be visited."""
    from types import SimpleNamespace

    import torch

    from torchdr_amd.distance import base as dbase

    def hopeless(dist, radius, tiles, tau):
        ci = SimpleNamespace(dist=dist, radius=radius, tiles=tiles)
        return dbase.ClusterIndex.tiles_hopeless(ci, tau)

    C = 64
    g = torch.Generator().manual_seed(0)
    tiles = torch.full((C,), 30, dtype=torch.int32)
    # structureless: centre distances ~3, radii ~11, k-th neighbour distance^2 ~ 150
    dist = 3.0 + torch.rand(C, C, generator=g)
    dist = (dist + dist.T) / 2
    dist.fill_diagonal_(0.0)
    assert hopeless(dist, torch.full((C,), 11.0), tiles, 150.0) is True
    # overlapping blobs that tiles still separate
    dist = torch.full((C, C), 16.0)
    dist.fill_diagonal_(0.0)
    assert hopeless(dist, torch.full((C,), 6.6), tiles, 64.0) is False
    # the same blobs with a threshold beyond the best bound: every tile would be visited
    assert hopeless(dist, torch.full((C,), 6.6), tiles, 10.8 ** 2) is True
    # one far outlier cluster of a single tile among overlapping ones: skipping it alone is not worth a table
    dist = 3.0 * torch.ones(C, C)
    dist[0, :] = dist[:, 0] = 200.0
    dist.fill_diagonal_(0.0)
    t2 = tiles.clone()
    t2[0] = 1
    assert hopeless(dist, torch.full((C,), 11.0), t2, 150.0) is True


def test_tier_cost_model_on_the_host():
    """`distance/base.py:_pick_tier` (no device): estimated time = scan + exact re-search of the rows a tier will flag.  On an
    unpruned scan a few per cent of flagged rows cost next to nothing against the scan, so the cheaper tier wins; on a pruned scan
    (a few per cent of the tiles visited) the same rows cost several scans, and the tier that flags none wins -- round 3's k = 15
    case at the headline size and round 5's N = 1M, D = 256 case (tier 1 at 4.9 % against the long-list tier)."""
    from torchdr_amd.distance import base as dbase

    n, d = 1_000_000, 128
    # unpruned: one term with 2 % flagged (20 ms of re-search) against three terms (0.4 s more scan)
    assert dbase._pick_tier([(0, 0.02, 50.0), (1, 0.0, 50.0)], n, n, d, None)[0] == 0
    # pruned to ~3 % of the tiles: the 20 ms are several scans
    assert dbase._pick_tier([(0, 0.02, 50.0), (1, 0.0, 50.0)], n, n, d, lambda tau: 0.03)[0] == 1
    # nothing flagged anywhere: the cheapest matrix work
    assert dbase._pick_tier([(0, 0.0, 50.0), (1, 0.0, 50.0)], n, n, d, lambda tau: 0.03)[0] == 0
    # D = 256: 4.9 % of 1M rows re-searched at 4 us each (200 ms) against the long-list tier's scan
    assert dbase._pick_tier([(1, 0.049, 900.0), (2, 0.0, 900.0)], n, n, 256, lambda tau: 0.05)[0] == 2
    # the rule that gives the long-list tier its pilot: predicted re-search of the best candidate beyond _LONG_TIER_PILOT_SEC
    pairs = float(n) * n * 256 * 2.0
    assert 0.049 * pairs / dbase._EXACT_RATE > dbase._LONG_TIER_PILOT_SEC > 0.0005 * float(n) * n * d * 2.0 / dbase._EXACT_RATE
