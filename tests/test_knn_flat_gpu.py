"""The unpruned two-stage search as a threshold scan (csrc/tdr_knn_flat.hip; round 5) -- the same results as the
list-keeping kernel, the one-stage exact kernel and the CPU oracle, bit for bit: reference distance/torch.py:82-122
(`pairwise_distances_torch` + `kmin`), the unstructured benchmark set of benchmarks/faiss/run_benchmark.py:143-146."""

import ctypes

import pytest
import torch

from tests.conftest import gmm

pytestmark = pytest.mark.gpu


def _search(X, k, metric="sqeuclidean", **opts):
    from torchdr_amd import config
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance import pairwise_distances

    with config.options(**opts):
        C, I = pairwise_distances(X, metric=metric, k=k, exclude_diag=True, return_indices=True)
    return C, I, dict(dbase.LAST_KNN)


@pytest.mark.parametrize("scale,d,k,metric", [(0.0, 128, 15, "sqeuclidean"), (2.0, 64, 30, "euclidean"), (1.0, 20, 10, "sqeuclidean"),
                                              (6.0, 128, 30, "sqeuclidean"), (0.0, 256, 30, "sqeuclidean"), (0.5, 200, 15, "euclidean")])
def test_flat_scan_equals_list_kernel_and_exact(scale, d, k, metric):
    """140k points (4375 tiles: just above the threshold scan's minimum), unpruned: threshold scan == list-keeping kernel ==
    one-stage kernel on distances AND indices; 256 sampled rows == the CPU oracle."""
    import oracle

    n = 140_000
    X = gmm(n, d, scale, seed=17).cuda()
    Cf, If, info_f = _search(X, k, metric, PRUNE_MODE="0", FLAT_SCAN=True)
    assert info_f["path"] == "screen" and info_f.get("flat_terms") in ((1,) if d > 128 else (1, 2, 3)), info_f   # 128 < d <= 256: one term only
    assert info_f["flagged"] <= n // 50, info_f       # the scan answered (a flagged row is recomputed exactly: equality alone proves nothing)
    Cl, Il, info_l = _search(X, k, metric, PRUNE_MODE="0", FLAT_SCAN=False)
    assert info_l.get("flat_terms") == 0
    assert torch.equal(If, Il) and torch.equal(Cf, Cl)
    Ce, Ie, _ = _search(X, k, metric, PRUNE_MODE="0", SCREEN_MODE="0")
    assert torch.equal(If, Ie) and torch.equal(Cf, Ce)
    rows = torch.arange(0, n, n // 256)[:256]
    Xc = X.cpu()
    for r in rows.tolist()[:64]:
        Co, Io = oracle.knn(Xc[r:r + 1], k, metric, True, Y=Xc, q_offset=r)
        assert torch.equal(If[r].cpu(), Io[0]) and torch.equal(Cf[r].cpu(), Co[0])


@pytest.mark.parametrize("n,k,data", [(500_000, 30, "gauss"), (250_000, 100, "randn"), (262_144, 64, "gauss")])
def test_flat_scan_pass_plan_keeps_the_buffers_from_overflowing(n, k, data):
    """The passes' growth factor follows k (a pass that takes a query from n seen rows to r n appends ~ (r - 1) k entries to a
    256-entry region): structureless data at sizes / k where a fixed plan overflowed -- N = 500k had a pass growing 7.6x (8 % of
    the queries lost and recomputed, profiles/r05_knn_flat_matrix.jsonl), k = 100 lost every query.  Few flagged rows, and the
    list-keeping kernel's results bit for bit."""
    torch.manual_seed(11)
    X = (torch.randn(n, 128) if data == "randn" else gmm(n, 128, 0.0, seed=11)).cuda()
    Cf, If, info = _search(X, k, FLAT_SCAN=True)
    assert info["path"] == "screen" and info.get("flat_terms") in (1, 2, 3), info
    assert info["flagged"] <= n // 200, info
    Cl, Il, info_l = _search(X, k, FLAT_SCAN=False)
    assert info_l.get("flat_terms") == 0
    assert torch.equal(If, Il) and torch.equal(Cf, Cl)


@pytest.mark.parametrize("terms", [1, 2, 3])
def test_flat_scan_every_tier_equals_exact(terms):
    """Each tier of the threshold scan forced in turn (h.h'; h.h' + h.l'; all three products) on the tie-rich mixture: the
    one-stage kernel's distances and indices bit for bit; the wider the band, the more rows are flagged, never the other way."""
    n, d, k = 200_000, 96, 20
    X = gmm(n, d, 3.0, seed=23).cuda()
    Cf, If, info = _search(X, k, PRUNE_MODE="0", FLAT_SCAN=True, FLAT_FORCE_TERMS=terms)
    assert info.get("flat_terms") == terms and info["flagged"] <= n // 4, info
    Ce, Ie, _ = _search(X, k, PRUNE_MODE="0", SCREEN_MODE="0")
    assert torch.equal(Cf, Ce) and torch.equal(If, Ie)
    if terms == 3:
        assert info["flagged"] <= n // 1000, info


def test_flat_scan_with_duplicates_and_flagged_rows():
    """Exact duplicates (more copies than a list holds) overflow the band of their queries: those rows are flagged and
    recomputed exactly; every row equals the one-stage kernel."""
    n, d, k = 140_000, 32, 12
    X = gmm(n, d, 1.5, seed=3)
    X[1000:1400] = X[999]            # 401 identical points: their band holds > 128 candidates
    X[5000:5040] = X[4999]           # 41 identical points: fits the lists
    X = X.cuda()
    Cf, If, info = _search(X, k, PRUNE_MODE="0", FLAT_SCAN=True)
    print(info)
    assert info.get("flat_terms") in (1, 2, 3) and 400 <= info["flagged"] <= n // 10, info
    Ce, Ie, _ = _search(X, k, PRUNE_MODE="0", SCREEN_MODE="0")
    assert torch.equal(Cf, Ce) and torch.equal(If, Ie)


def test_flat_stages_scan_and_select():
    """tdr_knn_flat_scan_f32 / tdr_knn_flat_select_f32 on their own: with tau = +inf for a few queries and a tiny capacity the
    count says what was met and `lost` is raised; with real thresholds the select returns the ascending L smallest of list +
    appended (checked against a sort on the host)."""
    from torchdr_amd import _lib
    from torchdr_amd.distance import base as dbase

    L = _lib.lib()
    n, d, k, LL, cap = 140_000, 64, 10, 32, 64
    X = gmm(n, d, 1.0, seed=5).cuda()
    P = dbase.PackedPoints(X)
    q16, y16, meta = dbase._screen_operands(P, P)
    nq = 4096
    tau = torch.full((nq,), 40.0, device="cuda")        # sq. distances inside a blob are ~ 2 * 64 * 0.25 = 32
    tau[:8] = float("inf")
    buf = torch.zeros((nq, cap), dtype=torch.int64, device="cuda")
    cnt = torch.zeros(nq, dtype=torch.int32, device="cuda")
    n_tiles = (n + 31) // 32
    for terms in (1, 2, 3):
        _lib.check(L.tdr_knn_flat_scan_f32(_lib.ptr(q16), nq, 0, _lib.ptr(y16), n, d, terms, 1, 100, 1101, 1, _lib.ptr(meta), _lib.ptr(tau),
                                           _lib.ptr(buf), _lib.ptr(cnt), cap, _lib.stream_ptr()), "scan")
        c = cnt.cpu()
        # tau = inf: every row of the 1001 tiles (minus the query itself) is met and counted -- a column the wavefront's buffer
        # cannot take is appended straight to the query's region
        assert bool((c[:8] >= 1001 * 32 - 1).all()), (terms, c[:8])
        # reference: exact squared distances of the same block, thresholded with slack for the screening error
        D = torch.cdist(X[:nq].double(), X[3200:35232].double()) ** 2
        lo = (D <= 40.0 - 0.5).sum(1).cpu()
        hi = (D <= 40.0 + 0.5).sum(1).cpu()
        assert bool((c[64:] >= lo[64:] - 1).all()) and bool((c[64:] <= hi[64:]).all()), terms
        # appended keys: value within the band of the exact distance of (query, row)
        b = buf.cpu()
        for qi in (64, 100, 4095):
            m = min(int(c[qi]), cap)
            if m == 0:
                continue
            keys = b[qi, :m]
            rows = (keys & 0xFFFFFFFF).long()
            bits = ((keys >> 32) & 0xFFFFFFFF)
            bits = torch.where(bits >= 0x80000000, bits & 0x7FFFFFFF, (~bits) & 0xFFFFFFFF).to(torch.int32)
            vals = bits.view(torch.float32)
            ref = ((X[qi].double() - X[rows.cuda()].double()) ** 2).sum(1).cpu()
            assert bool((rows >= 3200).all()) and bool((rows < 35232).all()) and bool((rows != qi).all())
            assert torch.allclose(vals.double(), ref, atol=0.5), terms
    # select: list (empty) + appended -> the LL smallest ascending
    lst = torch.zeros((nq, LL), dtype=torch.int64, device="cuda")
    tau2 = torch.empty(nq, device="cuda")
    lost = torch.zeros(nq, dtype=torch.int32, device="cuda")
    _lib.check(L.tdr_knn_flat_select_f32(_lib.ptr(lst), 0, _lib.ptr(buf), _lib.ptr(cnt), 1, cap, _lib.ptr(P.norms), _lib.ptr(meta), nq, d, k, LL,
                                         3, _lib.ptr(tau2), _lib.ptr(lost), _lib.stream_ptr()), "select")
    c, b, ls, lo_ = cnt.cpu(), buf.cpu(), lst.cpu(), lost.cpu()
    SENT = -0x7FFFFF00000001     # 0xFF800000FFFFFFFF as int64
    for qi in (0, 8, 64, 100, 2000, 4095):
        m = min(int(c[qi]), cap)
        assert int(lo_[qi]) == (1 if (int(c[qi]) > cap or int(c[qi]) < 0) else 0)
        if int(c[qi]) > cap or int(c[qi]) < 0:
            continue        # a lost query is recomputed exactly: its list is not used
        # keys compare as UNSIGNED 64-bit numbers
        want = sorted((int(x) & 0xFFFFFFFFFFFFFFFF) for x in b[qi, :m].tolist())[:LL]
        got = [int(x) & 0xFFFFFFFFFFFFFFFF for x in ls[qi].tolist()]
        assert got[:len(want)] == want
        assert all(g == 0xFF800000FFFFFFFF for g in got[len(want):])
    assert SENT < 0


def test_flat_scan_on_rows_sorted_by_class():
    """A block whose rows come sorted by class (200 classes of 800 rows; the 64 queries of a wavefront share their neighbours):
    the pilot and every pass take tiles from all over the database (position j of the visiting order = tile (j * stride) mod
    n_tiles), so the thresholds are not taken from one class, and a tile full of a wavefront's neighbours is appended without
    loss.  The scan answers most rows itself (measured: 80 %; a query that meets no row of its own class before the pass that
    brings most of them overflows its 256-entry region and is recomputed exactly -- with contiguous ranges 97 % were) and the
    result equals the exact search."""
    n, d, k = 160_000, 48, 15
    g = torch.Generator().manual_seed(7)
    centers = torch.randn(200, d, generator=g) * 2.0
    labels = torch.arange(n) // (n // 200)                    # rows 0..799 class 0, 800..1599 class 1, ...
    X = (centers[labels] + 0.5 * torch.randn(n, d, generator=g)).cuda()
    Cf, If, info = _search(X, k, PRUNE_MODE="0", FLAT_SCAN=True)
    print(info)
    assert info.get("flat_terms") in (1, 2, 3) and info["flagged"] <= n // 3, info
    Ce, Ie, _ = _search(X, k, PRUNE_MODE="0", SCREEN_MODE="0")
    assert torch.equal(Cf, Ce) and torch.equal(If, Ie)


@pytest.mark.parametrize("separate", [False, True])
def test_flat_scan_on_a_query_chunk_with_offset(separate):
    """The row-chunked search of a sharded fit (distance/base.py:183-206 in the reference: a rank's chunk against the full
    block, self excluded by GLOBAL index): queries = rows 3200 .. 43199 of the block, either as a row slice of the same packed
    block or as a separate block with its own image and q_offset -- threshold scan == one-stage kernel, bit for bit; a cross
    search without exclusion as well."""
    from torchdr_amd import config
    from torchdr_amd.distance import base as dbase

    n, d, k, q0, nq = 140_000, 40, 12, 3200, 40_000
    X = gmm(n, d, 1.0, seed=23).cuda()
    Yp = dbase.PackedPoints(X)

    def run(**opts):
        with config.options(PRUNE_MODE="0", **opts):
            if separate:
                Qp = dbase.PackedPoints(X[q0:q0 + nq].clone())
                out = dbase.knn_packed(Qp, Yp, k, "sqeuclidean", True, q_offset=q0)
            else:
                out = dbase.knn_packed(Yp, Yp, k, "sqeuclidean", True, q_rows=slice(q0, q0 + nq))
        return out, dict(dbase.LAST_KNN)

    (Cf, If), info = run(FLAT_SCAN=True)
    assert info.get("flat_terms") in (1, 2, 3) and info["flagged"] <= nq // 50, info
    (Ce, Ie), _ = run(SCREEN_MODE="0")
    assert torch.equal(If, Ie) and torch.equal(Cf, Ce)
    assert not bool((If == torch.arange(q0, q0 + nq, device="cuda", dtype=torch.int32)[:, None]).any())
    if separate:        # cross search, nothing excluded: a query's own row is its first neighbour
        with config.options(PRUNE_MODE="0", FLAT_SCAN=True):
            Qp = dbase.PackedPoints(X[q0:q0 + nq].clone())
            Cx, Ix = dbase.knn_packed(Qp, Yp, k, "euclidean", False)
            assert dbase.LAST_KNN.get("flat_terms") in (1, 2, 3)
        with config.options(PRUNE_MODE="0", SCREEN_MODE="0"):
            Cy, Iy = dbase.knn_packed(Qp, Yp, k, "euclidean", False)
        assert torch.equal(Ix, Iy) and torch.equal(Cx, Cy)
