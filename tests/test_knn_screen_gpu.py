"""Two-stage exact kNN (fp16-split screening + exact rescoring) vs the one-stage exact kernel: the two paths
must agree BIT FOR BIT (values and indices), on every data regime -- including the ones where the screening
band overflows and the flagged queries are recomputed by the one-stage kernel."""

import pytest
import torch

from tests.conftest import gmm

pytestmark = pytest.mark.gpu


def both_paths(X, k, metric, exclude, Y=None):
    from torchdr_amd.distance import base as dbase

    old = dbase.SCREEN_MODE
    try:
        dbase.SCREEN_MODE = "0"
        Xp = dbase.PackedPoints(X)
        Yp = Xp if Y is None else dbase.PackedPoints(Y)
        C0, I0 = dbase.knn_packed(Xp, Yp, k, metric, exclude)
        assert dbase.LAST_KNN["path"] == "exact"
        dbase.SCREEN_MODE = "force"
        Xp = dbase.PackedPoints(X)
        Yp = Xp if Y is None else dbase.PackedPoints(Y)
        C1, I1 = dbase.knn_packed(Xp, Yp, k, metric, exclude)
        assert dbase.LAST_KNN["path"] == "screen"
        flagged = dbase.LAST_KNN["flagged"]
    finally:
        dbase.SCREEN_MODE = old
    return (C0, I0), (C1, I1), flagged


@pytest.mark.parametrize("scale", [0.0, 2.0, 10.0])
@pytest.mark.parametrize("d,k", [(128, 30), (128, 15), (32, 30), (50, 10), (64, 45), (100, 90), (200, 30), (256, 15)])
def test_screen_equals_exact(d, k, scale):
    X = gmm(6000, d, scale, seed=11 + d + k).cuda()
    (C0, I0), (C1, I1), flagged = both_paths(X, k, "sqeuclidean", True)
    assert torch.equal(I0, I1), f"indices differ ({int((I0 != I1).sum())} entries, {flagged} flagged)"
    assert torch.equal(C0, C1)


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean"])
@pytest.mark.parametrize("exclude", [False, True])
def test_screen_bit_exact_vs_oracle(metric, exclude):
    """Directly against the CPU oracle (itself pinned bit-exactly by the reference's golden vectors)."""
    import oracle
    from torchdr_amd.distance import base as dbase

    for scale in (0.0, 2.0, 10.0):
        X = gmm(2048, 128, scale)
        old = dbase.SCREEN_MODE
        try:
            dbase.SCREEN_MODE = "force"
            Xp = dbase.PackedPoints(X.cuda())
            C, I = dbase.knn_packed(Xp, Xp, 30, metric, exclude)
            assert dbase.LAST_KNN["path"] == "screen"
        finally:
            dbase.SCREEN_MODE = old
        Co, Io = oracle.knn(X, 30, metric, exclude)
        assert torch.equal(I.cpu(), Io), f"scale {scale}: {(I.cpu() != Io).any(1).sum().item()} rows differ"
        assert torch.equal(C.cpu(), Co)


def test_screen_cross_and_ragged():
    X = gmm(4097, 96, 2.0, seed=3).cuda()
    Y = (gmm(7001, 96, 2.0, seed=4) * 3.0).cuda()  # different magnitude: the shared scale must cover both blocks
    (C0, I0), (C1, I1), _ = both_paths(X, 20, "sqeuclidean", False, Y=Y)
    assert torch.equal(I0, I1) and torch.equal(C0, C1)


def test_screen_overflow_falls_back_to_exact():
    """A large common offset makes ||x|| ||y|| huge relative to the neighbour distances: the worst-case band
    swallows the spare list slots, queries are flagged and recomputed exactly."""
    X = (gmm(5000, 64, 1.0, seed=8) + 300.0).cuda()
    (C0, I0), (C1, I1), flagged = both_paths(X, 30, "sqeuclidean", True)
    assert flagged > 0
    assert torch.equal(I0, I1) and torch.equal(C0, C1)


def test_screen_duplicates_and_tiny_values():
    base = gmm(3000, 48, 2.0, seed=5)
    X = torch.cat([base, base[:500], base[:100] * 1e-6]).cuda()  # exact duplicates (tied distances) + tiny rows
    (C0, I0), (C1, I1), _ = both_paths(X, 25, "sqeuclidean", True)
    assert torch.equal(I0, I1) and torch.equal(C0, C1)


def test_screen_split_launch_small_query_count():
    """Few queries against a large database: the database is sliced over gridDim.y and the rescoring stage
    merges the per-slice candidate lists."""
    Y = gmm(200_000, 128, 2.0, seed=6).cuda()
    X = Y[:1000].contiguous()
    (C0, I0), (C1, I1), _ = both_paths(X, 30, "sqeuclidean", False, Y=Y)
    assert torch.equal(I0, I1) and torch.equal(C0, C1)


def test_screen_pilot_routes_unsuitable_data_to_one_stage_kernel():
    """Large search on data whose band overflows: the pilot slice notices and the whole search runs on the
    one-stage kernel (same results, no wasted screening pass)."""
    from torchdr_amd.distance import base as dbase

    X = (gmm(40000, 64, 1.0, seed=12) + 300.0).cuda()
    (C0, I0), (C1, I1), flagged = both_paths_allow_pilot(X, 15)
    assert torch.equal(I0, I1) and torch.equal(C0, C1)


def both_paths_allow_pilot(X, k):
    from torchdr_amd.distance import base as dbase

    old = dbase.SCREEN_MODE
    try:
        dbase.SCREEN_MODE = "0"
        Xp = dbase.PackedPoints(X)
        r0 = dbase.knn_packed(Xp, Xp, k, "sqeuclidean", True)
        dbase.SCREEN_MODE = "force"
        Xp = dbase.PackedPoints(X)
        r1 = dbase.knn_packed(Xp, Xp, k, "sqeuclidean", True)
        assert dbase.LAST_KNN["path"] == "exact (pilot overflow)"
    finally:
        dbase.SCREEN_MODE = old
    return r0, r1, 0


@pytest.mark.parametrize("scale,k", [(2.0, 30), (10.0, 30), (10.0, 50)])
def test_screen_long_list_tier_equals_exact(scale, k):
    """tier 2 (one workgroup per CU, up to k + 72 list slots, two entries per lane) against the one-stage kernel."""
    from torchdr_amd.distance import base as dbase

    X = gmm(6000, 128, scale, seed=21).cuda()
    old = dbase.SCREEN_MODE
    try:
        dbase.SCREEN_MODE = "0"
        Xp = dbase.PackedPoints(X)
        C0, I0 = dbase.knn_packed(Xp, Xp, k, "sqeuclidean", True)
    finally:
        dbase.SCREEN_MODE = old
    C1 = torch.empty_like(C0)
    I1 = torch.empty_like(I0)
    dbase._knn_screen(Xp, Xp, 0, Xp.n, k, "sqeuclidean", True, 0, C1, I1, pilot=False, tier=2)
    assert torch.equal(I0, I1) and torch.equal(C0, C1)


@pytest.mark.parametrize("scale", [0.0, 2.0, 10.0])
@pytest.mark.parametrize("d,k", [(128, 30), (64, 15), (100, 40)])
def test_screen_one_term_tier_equals_exact(d, k, scale):
    """tier 0 (h.h' only: a third of the matrix work, a 2^-10 |x||y| band, 62-entry lists): whatever overflows is
    recomputed by the one-stage kernel, so the result is still bit-identical."""
    from torchdr_amd.distance import base as dbase

    X = gmm(6000, d, scale, seed=31 + d).cuda()
    old = dbase.SCREEN_MODE
    try:
        dbase.SCREEN_MODE = "0"
        Xp = dbase.PackedPoints(X)
        C0, I0 = dbase.knn_packed(Xp, Xp, k, "sqeuclidean", True)
    finally:
        dbase.SCREEN_MODE = old
    C1 = torch.empty_like(C0)
    I1 = torch.empty_like(I0)
    dbase._knn_screen(Xp, Xp, 0, Xp.n, k, "sqeuclidean", True, 0, C1, I1, pilot=False, tier=0)
    assert torch.equal(I0, I1) and torch.equal(C0, C1)


@pytest.mark.parametrize("scale,n,d,k,tier", [(2.0, 20000, 128, 30, 1), (0.0, 9000, 64, 15, 0), (2.0, 7001, 100, 40, 1),
                                              (4.0, 30000, 32, 10, 1), (2.0, 12000, 256, 30, 1), (1.0, 8000, 192, 20, 0)])
def test_screen_cluster_pruned_scan_equals_exact(scale, n, d, k, tier):
    """Cluster-bound pruning (points sorted by a coarse k-means, clusters padded to tile boundaries, clusters whose
    ball cannot reach the thresholds skipped): results must not depend on it -- bit-identical to the one-stage kernel,
    in the caller's row order and index space, ties broken by the caller's indices."""
    from torchdr_amd.distance import base as dbase

    base = gmm(n, d, scale, seed=41 + d)
    X = torch.cat([base, base[:300]]).cuda()  # exact duplicates: tied distances across different clusters' tiles
    old = (dbase.SCREEN_MODE, dbase.PRUNE_MODE)
    try:
        dbase.SCREEN_MODE, dbase.PRUNE_MODE = "0", "0"
        Xp = dbase.PackedPoints(X)
        C0, I0 = dbase.knn_packed(Xp, Xp, k, "sqeuclidean", True)
        dbase.PRUNE_MODE = "force"
        C1 = torch.empty_like(C0)
        I1 = torch.empty_like(I0)
        dbase._knn_screen(Xp, Xp, 0, Xp.n, k, "sqeuclidean", True, 0, C1, I1, pilot=False, tier=tier)
        assert dbase.LAST_KNN["pruned"]
    finally:
        dbase.SCREEN_MODE, dbase.PRUNE_MODE = old
    assert torch.equal(I0, I1), f"{int((I0 != I1).any(1).sum())} rows differ"
    assert torch.equal(C0, C1)


@pytest.mark.parametrize("scale,n,d,k,tier,dups", [(2.0, 40000, 128, 30, 1, 0), (2.0, 40000, 128, 60, 1, 0), (0.7, 30000, 64, 15, 0, 0),
                                                   (3.0, 25000, 256, 30, 1, 0), (2.0, 30000, 128, 30, 1, 400), (5.0, 6000, 16, 5, 1, 0),
                                                   (2.0, 20000, 100, 70, 1, 0)])
def test_pruned_scan_lazy_buffers_equal_sorted_lists_and_exact(scale, n, d, k, tier, dups):
    """The exact cluster-pruned search keeps UNSORTED per-query candidate buffers, compacted when full (round 6,
    tdr_knn_screen_clustered_lists(1), the default), instead of the sorted lists of rounds 2-5 (0): both give the one-stage
    kernel's rows bit for bit -- tight blobs (nearly every tile of a query's own cluster holds survivors), overlapping blobs,
    k above what the buffers serve (the sorted lists take over), D = 256 (shorter buffers), tiny k, and `dups` copies of one
    point (more candidates inside the error band than a buffer holds: those queries are reported lost and recomputed exactly)."""
    from torchdr_amd import _lib
    from torchdr_amd.distance import base as dbase

    L = _lib.lib()
    base = gmm(n, d, scale, seed=5 + d + k)
    parts = [base, base[:300]]
    if dups:
        parts.append(base[1234:1235].expand(dups, d))
    X = torch.cat(parts).cuda()
    old = (dbase.SCREEN_MODE, dbase.PRUNE_MODE)
    res = {}
    try:
        dbase.SCREEN_MODE, dbase.PRUNE_MODE = "0", "0"
        Xp = dbase.PackedPoints(X)
        C0, I0 = dbase.knn_packed(Xp, Xp, k, "sqeuclidean", True)
        dbase.PRUNE_MODE = "force"
        for mode in (1, 0):
            prev = L.tdr_knn_screen_clustered_lists(mode)
            try:
                C1 = torch.empty_like(C0)
                I1 = torch.empty_like(I0)
                bad = dbase._knn_screen(Xp, Xp, 0, Xp.n, k, "sqeuclidean", True, 0, C1, I1, pilot=False, tier=tier)
                assert dbase.LAST_KNN["pruned"]
                res[mode] = (C1, I1, bad)
            finally:
                L.tdr_knn_screen_clustered_lists(prev)
    finally:
        dbase.SCREEN_MODE, dbase.PRUNE_MODE = old
    assert L.tdr_knn_screen_clustered_lists(1) == 1      # the default
    for mode in (1, 0):
        C1, I1, bad = res[mode]
        assert torch.equal(I0, I1), f"lists mode {mode}: {int((I0 != I1).any(1).sum())} rows differ"
        assert torch.equal(C0, C1)
    if dups:
        assert res[1][2] >= dups      # every copy has the other copies at distance 0 inside its band


def test_pruned_search_picks_its_list_form_by_the_predicted_scan_share():
    """Default dispatch (pilot -> tier -> cluster-pruned search): separated blobs are predicted to visit a per mille of the tiles
    and take the lazy buffers; `config.options(PRUNED_LISTS=...)` forces either form; all three return the same rows."""
    from torchdr_amd import config
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance import pairwise_distances

    X = gmm(200_000, 64, 2.0, seed=3).cuda()
    C0, I0 = pairwise_distances(X, metric="sqeuclidean", k=20, exclude_diag=True, return_indices=True)
    assert dbase.LAST_KNN["path"] == "screen-pruned" and dbase.LAST_KNN["lists"] == "lazy"
    assert dbase.LAST_KNN["predicted_share"] is not None and dbase.LAST_KNN["predicted_share"] <= dbase._LAZY_MAX_SHARE
    for want in ("sorted", "lazy"):
        with config.options(PRUNED_LISTS=want):
            C1, I1 = pairwise_distances(X, metric="sqeuclidean", k=20, exclude_diag=True, return_indices=True)
        assert dbase.LAST_KNN["path"] == "screen-pruned" and dbase.LAST_KNN["lists"] == want
        assert torch.equal(C0, C1) and torch.equal(I0, I1)


def test_index_refinement_and_robust_pilot_threshold_on_heavy_tailed_group_sizes():
    """More separated groups than the first seeding finds (Zipf-sized groups, the small ones absent from the 8192-point sample): every
    stray group sets the radius of the ball that absorbs it and rows of groups with fewer than k members have far k-th neighbours.
    The index is re-seeded on the points far from every centre (ClusterIndex._refine) and the predictions are made for the pilot's
    90th-percentile k-th distance (`_pilot_tau`): the search is pruned -- and returns the rows of the unrefined, unpruned search."""
    from torchdr_amd import config
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance import pairwise_distances

    n, d, k, groups = 300_000, 128, 30, 2500
    g = torch.Generator().manual_seed(1)
    w = 1.0 / torch.arange(1, groups + 1, dtype=torch.float64) ** 1.1
    lab = torch.multinomial(w / w.sum(), n, replacement=True, generator=g)
    X = (torch.randn(groups, d, generator=g)[lab] * 2.0 + 0.5 * torch.randn(n, d, generator=g)).float().cuda().contiguous()
    dbase.LAST_KNN.pop("index_refined", None)
    C0, I0 = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
    L = dict(dbase.LAST_KNN)
    assert L["path"] == "screen-pruned" and L.get("index_refined"), L
    assert L["pilot_tau"][0] == "q90" and L["pilot_tau"][1] > 4 * L["pilot_tau"][2]
    first, last = L["index_refined"][0], L["index_refined"][-1]
    assert last[1] > first[0] and last[1] <= 4096           # balls were added, within the tables' limit
    with config.options(REFINE_INDEX=False, PRUNE_MODE="0"):
        C1, I1 = pairwise_distances(X.clone(), metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
    assert dbase.LAST_KNN["path"] != "screen-pruned"
    assert torch.equal(C0, C1) and torch.equal(I0, I1)
    # the benchmark's mixture never triggers either mechanism (equal groups, homogeneous radii)
    dbase.LAST_KNN.pop("index_refined", None)
    pairwise_distances(gmm(200_000, 64, 2.0, seed=3).cuda(), metric="sqeuclidean", k=20, exclude_diag=True)
    assert not dbase.LAST_KNN.get("index_refined") and dbase.LAST_KNN["pilot_tau"][0] == "max"


def test_headline_size_search_sampled_against_the_one_stage_kernel():
    """BASELINE's full size (N = 1M, D = 128, k = 30), default dispatch (pilot -> tier -> cluster-pruned two-stage search):
    256 sampled rows searched by the CPU oracle against the whole set, and 8192 sampled rows re-searched by the one-stage exact fp32 kernel -- itself bit-exact against the CPU oracle at the
    sizes the oracle finishes -- give the same neighbours and distances bit for bit; every row is sorted and never
    returns itself; the neighbour relation is the kNN graph of a metric (i in N(j) => d_ij <= d_j,k)."""
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance import pairwise_distances

    n, k = 1_000_000, 30
    Xh = gmm(n, 128, 2.0)
    X = Xh.cuda()
    C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
    assert dbase.LAST_KNN["path"] == "screen-pruned" and dbase.LAST_KNN["flagged"] == 0
    # the CPU oracle itself at full size: 256 sampled rows against all 1M points (top k + 1 without exclusion, own row
    # dropped) -- distances and indices bit for bit
    import oracle

    orows = torch.randint(0, n, (256,), generator=torch.Generator().manual_seed(11))
    Co, Io = oracle.knn(Xh[orows].contiguous(), k + 1, "sqeuclidean", exclude_self=False, Y=Xh)
    keep_o = Io != orows[:, None].int()
    ok_o = keep_o.sum(1) == k
    assert int(ok_o.sum()) >= 250
    assert torch.equal(Io[ok_o][keep_o[ok_o]].reshape(-1, k), I[orows.cuda()][ok_o.cuda()].cpu())
    assert torch.equal(Co[ok_o][keep_o[ok_o]].reshape(-1, k), C[orows.cuda()][ok_o.cuda()].cpu())
    assert bool((C[:, 1:] >= C[:, :-1]).all())
    assert not bool((I == torch.arange(n, device="cuda", dtype=torch.int32)[:, None]).any())
    rows = torch.randint(0, n, (8192,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    old = dbase.SCREEN_MODE
    dbase.SCREEN_MODE = "0"
    try:
        Ce, Ie = pairwise_distances(X[rows].contiguous(), X, metric="sqeuclidean", k=k + 1, return_indices=True)
    finally:
        dbase.SCREEN_MODE = old
    assert dbase.LAST_KNN["path"] == "exact"
    keep = Ie != rows[:, None].int()
    ok = keep.sum(1) == k                                   # the row itself is among its own k + 1 nearest (no duplicates)
    assert float(ok.float().mean()) > 0.999
    assert torch.equal(Ie[ok][keep[ok]].reshape(-1, k), I[rows][ok])
    assert torch.equal(Ce[ok][keep[ok]].reshape(-1, k), C[rows][ok])
    # consistency of the graph with the distances: if j lists i, then d(i, j) <= j's k-th distance, so either i lists j or
    # i's own k-th distance is smaller still
    j = I[rows].long()                                      # (m, k) neighbours of the sampled rows
    dij = C[rows]
    back = (I[j.reshape(-1)].reshape(len(rows), k, k) == rows[:, None, None].int()).any(2)
    assert bool((back | (C[j.reshape(-1), k - 1].reshape(len(rows), k) <= dij)).all())


def test_long_list_tier_competes_when_flagged_rows_would_be_costly():
    """N = 1M, D = 256, centre scale 5, k = 15 (`profiles/r05_knn_pruned_matrix.jsonl`): the three-term tier passes its 512-query
    pilot just below 5 % flagged, and every flagged row costs a full one-stage scan of a 1M x 256 database -- 6.4 % of the rows were
    recomputed, 322 ms instead of 69.  The long-list tier's pilot now runs whenever the best candidate's predicted re-search is worth
    more than a pilot launch, and the cost model picks between them: few flagged rows, results checked on sampled rows against the
    one-stage kernel."""
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance import pairwise_distances

    n, d, k = 1_000_000, 256, 15
    X = gmm(n, d, 5.0).cuda()
    C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
    info = dict(dbase.LAST_KNN)
    assert info["path"] == "screen-pruned" and info["flagged"] <= n // 100, info
    rows = torch.randint(0, n, (2048,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    old = dbase.SCREEN_MODE
    dbase.SCREEN_MODE = "0"
    try:
        Ce, Ie = pairwise_distances(X[rows].contiguous(), X, metric="sqeuclidean", k=k + 1, return_indices=True)
    finally:
        dbase.SCREEN_MODE = old
    keep = Ie != rows[:, None].int()
    ok = keep.sum(1) == k
    assert float(ok.float().mean()) > 0.999
    assert torch.equal(Ie[ok][keep[ok]].reshape(-1, k), I[rows][ok])
    assert torch.equal(Ce[ok][keep[ok]].reshape(-1, k), C[rows][ok])


def test_cluster_index_is_the_same_on_every_run():
    """The Lloyd update sums a cluster's members in sample order (no atomics): centres equal the per-label means, and two
    builds of the index give the same labels bit for bit -- callers renumber their points by this order (UMAP's loop)."""
    from torchdr_amd import _lib
    from torchdr_amd.distance.base import ClusterIndex, PackedPoints

    L = _lib.lib()
    gen = torch.Generator().manual_seed(4)
    S, d, C = 5000, 48, 37
    Xs = torch.randn(S, d, generator=gen).cuda()
    labels = torch.randint(0, C - 1, (S,), generator=gen).to(torch.int32).cuda()   # cluster C - 1 stays empty
    cent0 = torch.randn(C, d, generator=gen).cuda()
    cent = cent0.clone()
    ws = torch.empty(C * d + 3 * C, dtype=torch.int32, device="cuda")
    _lib.check(L.tdr_cluster_update_f32(_lib.ptr(Xs), S, d, _lib.ptr(labels), C, _lib.ptr(cent), _lib.ptr(ws), _lib.stream_ptr()), "update")
    ref = torch.stack([Xs[labels == c].double().mean(0) for c in range(C - 1)])
    assert torch.allclose(cent[:-1].double(), ref, rtol=1e-5, atol=1e-6)
    assert torch.equal(cent[-1], cent0[-1])
    cent2 = cent0.clone()
    _lib.check(L.tdr_cluster_update_f32(_lib.ptr(Xs), S, d, _lib.ptr(labels), C, _lib.ptr(cent2), _lib.ptr(ws), _lib.stream_ptr()), "update")
    assert torch.equal(cent, cent2)
    X = gmm(60000, 32, 2.0, seed=12).cuda()
    a, b = ClusterIndex(PackedPoints(X)), ClusterIndex(PackedPoints(X))
    assert a.n_img == b.n_img and torch.equal(a.tile_cluster, b.tile_cluster) and torch.equal(a.radius, b.radius)
    assert torch.equal(a.row_map.sort().values, b.row_map.sort().values)
    # the predicted scan share (one launch, exact integer sums: tdr_cluster_scan_fraction_f32) against the tensor formulation it
    # replaced, over thresholds from "own cluster only" to "everything"; asked twice, the second answer comes from the memo
    t = a.tiles.to(torch.float32)
    seen = []
    for tau in (0.0, 0.5, 4.0, 30.0, 200.0, 1e9):
        gap = (a.dist - a.radius[:, None] - a.radius[None, :]).clamp_(min=0)
        ref = float((torch.mv((gap * gap <= tau).float(), t) * t).sum() / (t.sum() ** 2))
        got = a.scan_fraction(tau)
        assert abs(got - ref) < 1e-5, (tau, got, ref)
        assert a.scan_fraction(tau) == got and b.scan_fraction(tau) == got
        seen.append(got)
    assert seen == sorted(seen) and seen[-1] == 1.0 and seen[0] > 0.0


@pytest.mark.parametrize("S,C,d,blobs", [(2000, 200, 16, 0), (8000, 1000, 32, 1000), (16384, 2048, 8, 300), (777, 777, 4, 0), (1500, 400, 2, -1)])
def test_farthest_point_seeding_equals_the_greedy_selection(S, C, d, blobs):
    """tdr_cluster_maxmin_f32 / _adaptive_f32 -- two seeds per dependent step (round 6, the default) and the one-chain kernel of
    rounds 2-5 -- against each other and against a greedy farthest-point selection in numpy on the same fp32 matrix (ties: smallest
    index), incl. a sample as large as the kernel's capacity and C = S (every point; duplicates at distance 0); the adaptive form
    returns a prefix of the fixed form's seeds and stops at the same count in both kernels."""
    import numpy as np

    from torchdr_amd import _lib

    L = _lib.lib()
    gen = torch.Generator().manual_seed(S + C)
    if blobs < 0:
        X = torch.randint(0, 12, (S, d), generator=gen).float()     # 144 distinct points: every max-min distance is 0 after 144 seeds
    elif blobs:
        X = torch.randn(blobs, d, generator=gen)[torch.randint(0, blobs, (S,), generator=gen)] * 6 + torch.randn(S, d, generator=gen) * 0.3
    else:
        X = torch.randn(S, d, generator=gen)
    D2 = torch.cdist(X.double(), X.double()).pow(2).float().cuda().contiguous()
    assert S <= L.tdr_cluster_maxmin_capacity()
    assert L.tdr_cluster_maxmin_mode(1) == 1     # two seeds per dependent step is the default
    runs = {}
    for mode in (1, 0):
        prev = L.tdr_cluster_maxmin_mode(mode)
        try:
            seeds = torch.full((C,), -1, dtype=torch.int32, device="cuda")
            _lib.check(L.tdr_cluster_maxmin_f32(_lib.ptr(D2), D2.stride(0), S, C, _lib.ptr(seeds), _lib.stream_ptr()), "maxmin")
            ad = torch.full((C,), -1, dtype=torch.int32, device="cuda")
            n = torch.zeros(1, dtype=torch.int32, device="cuda")
            for drop in (0.25, 0.9):
                _lib.check(L.tdr_cluster_maxmin_adaptive_f32(_lib.ptr(D2), D2.stride(0), S, max(C // 4, 1), C, drop, _lib.ptr(ad), _lib.ptr(n),
                                                             _lib.stream_ptr()), "maxmin_adaptive")
                na = int(n.item())
                assert max(C // 4, 1) <= na <= C and torch.equal(ad.cpu()[:na], seeds.cpu()[:na])
                runs[(mode, drop)] = na
            runs[mode] = seeds.cpu()
        finally:
            L.tdr_cluster_maxmin_mode(prev)
    assert torch.equal(runs[1], runs[0]) and runs[(1, 0.25)] == runs[(0, 0.25)] and runs[(1, 0.9)] == runs[(0, 0.9)]
    got = runs[1]
    Dn = D2.cpu().numpy()
    mind = np.full(S, 3.0e38, dtype=np.float32)
    cur, ref = 0, []
    for _ in range(min(C, 300)):
        ref.append(cur)
        mind = np.minimum(mind, Dn[cur])
        cur = int(np.argmax(np.maximum(mind, 0)))       # first (smallest) index among equal maxima
    assert got[: len(ref)].tolist() == ref


def test_cluster_index_reads_the_number_of_groups_off_the_data():
    """No cluster count asked for: the farthest-point seeding goes on past the default (N / 1000) and stops where the max-min
    distance collapses -- one ball per well-separated group.  200k points in 700 blobs: 700 balls, not the default 200 (whose
    balls each swallow three or four blobs, so that nothing can be pruned); the pruned search then runs and is bit-equal to
    the exact one.  Data without groups (one Gaussian): the default count."""
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance.base import ClusterIndex, PackedPoints

    n, d, blobs = 200_000, 64, 700
    g = torch.Generator().manual_seed(21)
    centres = torch.randn(blobs, d, generator=g) * 2.0
    X = (centres[torch.arange(n) % blobs] + 0.5 * torch.randn(n, d, generator=g)).cuda()
    ci = ClusterIndex(PackedPoints(X))
    assert ci.n_clusters == blobs
    assert ClusterIndex(PackedPoints(X), n_clusters=300).n_clusters == 300          # an explicit count is taken as it is
    flat = torch.randn(n, d, generator=g).cuda()
    assert ClusterIndex(PackedPoints(flat)).n_clusters == n // 1000
    C, I = dbase.pairwise_distances(X, metric="sqeuclidean", k=15, exclude_diag=True, return_indices=True)
    assert dbase.LAST_KNN["path"].endswith("pruned"), dbase.LAST_KNN
    rows = torch.arange(0, n, 401, device="cuda")
    P = PackedPoints(X)
    Ce, Ie = dbase.knn_packed(PackedPoints(X[rows].contiguous()), P, 16, "sqeuclidean", False, _allow_screen=False)
    keep = Ie != rows[:, None].int()
    ok = keep.sum(1) == 15
    assert float(ok.float().mean()) > 0.999
    assert torch.equal(Ie[ok][keep[ok]].reshape(-1, 15), I[rows][ok])
    assert torch.equal(Ce[ok][keep[ok]].reshape(-1, 15), C[rows][ok])


@pytest.mark.parametrize("n,d,scale,k", [(120_000, 64, 1.0, 15), (150_000, 128, 1.1, 30)])
def test_tile_bounds_prune_overlapping_balls_and_equal_exact(n, d, scale, k):
    """Blobs whose balls overlap (centre distance ~ 1.4 sqrt(d), radius ~ 0.6 sqrt(d)): the ball-to-ball bound keeps every
    cluster, the per-tile table (|x - c| - R_c from the tile's own rows) does not.  The table is a LOWER bound of every row's
    distance to every centre; the search with it returns the same rows as the exact kernel."""
    from torchdr_amd import config
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance.base import PackedPoints

    X = gmm(n, d, scale, seed=5).cuda()
    with config.options(TILE_BOUNDS="force"):
        P = PackedPoints(X)
        C, I = dbase.knn_packed(P, P, k, "sqeuclidean", True)
    assert dbase.LAST_KNN["tile_bounds"] and dbase.LAST_KNN["pruned"], dbase.LAST_KNN
    ci = P._cluster_index
    T = ci.tile_cdist
    # the table against float64 distances of sampled tiles
    tiles = torch.arange(0, T.shape[0], max(T.shape[0] // 40, 1), device="cuda")
    cent = ci.centres.double()
    for t in tiles.tolist():
        rows = ci.row_map[t * 32:(t + 1) * 32]
        rows = rows[rows >= 0].long()
        if rows.numel() == 0:
            assert bool(torch.isinf(T[t]).all())
            continue
        true_min = torch.cdist(X[rows].double(), cent).min(0).values
        assert bool((T[t].double() <= true_min * (1 + 1e-7)).all())
        assert bool((T[t].double() >= true_min * (1 - 1e-3) - 1e-3).all())      # and not needlessly loose
    # results: sampled rows against the one-stage kernel
    rows = torch.arange(0, n, 157, device="cuda")
    Ce, Ie = dbase.knn_packed(PackedPoints(X[rows].contiguous()), PackedPoints(X), k + 1, "sqeuclidean", False, _allow_screen=False)
    keep = Ie != rows[:, None].int()
    ok = keep.sum(1) == k
    assert float(ok.float().mean()) > 0.999
    assert torch.equal(Ie[ok][keep[ok]].reshape(-1, k), I[rows][ok])
    assert torch.equal(Ce[ok][keep[ok]].reshape(-1, k), C[rows][ok])
    # and the whole result against the search without the table
    with config.options(TILE_BOUNDS=False):
        C0, I0 = dbase.knn_packed(PackedPoints(X), PackedPoints(X), k, "sqeuclidean", True) if False else dbase.pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
    assert torch.equal(C0, C) and torch.equal(I0, I)
