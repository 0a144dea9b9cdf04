"""Pairwise distances / exact kNN on MI355X -- drop-in for ``torchdr.distance``.

Mirrors the reference's function surface (``/root/reference/torchdr/distance/base.py:22-249``
``pairwise_distances`` and ``:252-405`` ``pairwise_distances_indexed``): same argument names,
meaning and error messages.  ``backend`` is accepted for signature compatibility; every backend
value ("faiss", "keops", None, a FaissConfig-like object) maps onto the one exact HIP kernel
(K1, ``csrc/tdr_knn.hip``) -- exact brute-force search is what Faiss ``Flat`` and the torch
path both compute.
"""

from typing import Optional

import torch

from torchdr_amd import _lib
from torchdr_amd.distributed import DistributedContext
from torchdr_amd.utils.dataloader import is_dataloader
from torchdr_amd.utils.misc import as_float32

LIST_METRICS = ["euclidean", "sqeuclidean", "manhattan", "angular", "sqhyperbolic"]

def _opt(name):
    """A behaviour switch of this module: the scoped override (torchdr_amd.config.options) or the module attribute."""
    from torchdr_amd import config

    return config.get(name, globals())

_METRIC_ID = {"sqeuclidean": 0, "euclidean": 1, "angular": 2, "manhattan": 3, "sqhyperbolic": 4}
_GENERAL_ONLY = ("manhattan", "sqhyperbolic")     # metrics that never take the MFMA scan kernels
PILOT_CONCURRENT = True   # pilot tiers and the cluster-index build on concurrent streams (False: one after the other)
WIDE_SCAN = True   # D > 256: K-chunked MFMA scan (False = library GEMM + top-k merge, kept for comparison)

# Value the reference adds to the diagonal when exclude_diag=True (distance/torch.py:115).
_DIAG_ADD = 1e12

# bench.py sets this to a list to collect (start_event, end_event, n_queries, kernel) around every scan launch
# (HIP events on the launch stream); None = no instrumentation.
PROFILE = None

# Two-stage search (fp16-split screening + exact rescoring, csrc/tdr_knn_screen.hip): "auto" uses it when
# the problem is large enough to pay for the extra packing pass, "0" never, "force" whenever it is supported.
import os as _os

SCREEN_MODE = _os.environ.get("TDR_KNN_SCREEN", "auto")
_SCREEN_MIN_PAIRS = 1 << 26  # nq * n_db below which the one-stage kernel is used
_SCREEN_PILOT_MIN_Q = 32768   # searches with at least this many queries screen a pilot slice first
_SCREEN_PILOT_Q = 512    # 4 query groups x 32 database slices = 128 workgroups per tier: the two concurrent tier pilots fill the chip together (1024 queries: 256 workgroups each, 3.2 ms alone, ~6 ms side by side)
_SCREEN_PILOT_MAX_FRAC = 0.05  # overflowed share of the pilot above which the one-stage kernel is used
# counters of the last knn_packed call (tests / bench): path taken and number of overflowed queries
LAST_KNN = {"path": None, "flagged": 0}


class PackedPoints:
    """A point block rewritten into MFMA tile images (+ squared norms) on the device.

    Packing costs one pass over the N x D block; callers that search the same database
    repeatedly (row-chunked / multi-GPU search) keep the object and reuse it.
    """

    def __init__(self, X: torch.Tensor):
        _lib.require_gpu(X, "X")
        if X.dim() != 2:
            raise ValueError("[TorchDR] ERROR : input must be 2-D (n_samples, n_features).")
        if X.dtype != torch.float32:
            raise NotImplementedError(
                "[torchdr_amd] only float32 inputs are supported by the HIP distance kernels "
                f"(got {X.dtype})."
            )
        if X.stride(1) != 1:
            X = X.contiguous()
        L = _lib.lib()
        self.n, self.d = X.shape
        nfl = L.tdr_packed_floats(self.n, self.d)
        if nfl == 0:
            raise NotImplementedError(
                f"[torchdr_amd] feature dimension {self.d} > 256 is not supported by the MFMA kNN kernel yet."
            )
        self.data = torch.empty(nfl, dtype=torch.float32, device=X.device)
        self.norms = torch.empty(self.n, dtype=torch.float32, device=X.device)
        _lib.check(
            L.tdr_pack_rows_f32(
                _lib.ptr(X), self.n, self.d, X.stride(0), _lib.ptr(self.data), _lib.ptr(self.norms),
                _lib.stream_ptr(),
            ),
            "tdr_pack_rows_f32",
        )
        self.device = X.device
        self.X = X  # row-major source (kept alive: the rescoring stage of the two-stage search reads it)
        self._img16 = None
        self._meta = None

    @classmethod
    def from_batches(cls, batches, n: int, d: int, device, dtype=torch.float32):
        """Point block assembled from a stream of row batches (DataLoader input, distance/faiss.py:477-591 streams its
        batches into the index the same way): every batch is copied into its rows of the resident block and the tiles it
        completes are packed at once (``tdr_pack_rows_f32`` on the tile range, enqueued while the loader prepares the next
        batch) -- when the last batch has arrived the tile images and norms exist; no second pass over the N x D block.
        ``n`` = upper bound of the row count (``len(dataset)``); a loader that yields fewer rows (``drop_last``) gives a
        shorter block.  Same images, bit for bit, as ``PackedPoints(X)`` of the concatenated batches."""
        L = _lib.lib()
        nfl = L.tdr_packed_floats(n, d)
        if nfl == 0 or dtype != torch.float32:
            return None
        self = cls.__new__(cls)
        X = torch.empty((n, d), dtype=torch.float32, device=device)
        self.data = torch.empty(nfl, dtype=torch.float32, device=device)
        self.norms = torch.empty(n, dtype=torch.float32, device=device)
        tile = int(L.tdr_packed_floats(32, d))
        pos = done = 0

        def pack(r0, r1):
            _lib.check(L.tdr_pack_rows_f32(_lib.ptr(X[r0:]), r1 - r0, d, X.stride(0), _lib.ptr(self.data[(r0 // 32) * tile:]),
                                           _lib.ptr(self.norms[r0:]), _lib.stream_ptr()), "tdr_pack_rows_f32")

        for b in batches:
            m = b.shape[0]
            if b.shape[1] != d:
                raise ValueError(f"[TorchDR] DataLoader batches disagree on the number of features ({b.shape[1]} vs {d}).")
            if pos + m > n:
                raise ValueError(f"[TorchDR] DataLoader yielded more than len(dataset) = {n} samples.")
            X[pos:pos + m].copy_(b, non_blocking=True)
            pos += m
            full = (pos // 32) * 32
            if full > done:
                pack(done, full)
                done = full
        if pos > done:
            pack(done, pos)
        if pos == 0:
            raise ValueError("[TorchDR] DataLoader is empty, cannot determine metadata. Ensure DataLoader yields at least one batch.")
        self.n, self.d = pos, d
        self.X = X if pos == n else X[:pos]
        self.norms = self.norms[:pos]
        self.device = X.device
        self._img16 = None
        self._meta = None
        return self

    def screen_image(self, meta: Optional[torch.Tensor] = None):
        """fp16-split tile images for the screening stage.  ``meta`` (2 x int32 device tensor) carries the
        shared scale of a query/database pair; without it the block's own maximum is used and cached."""
        L = _lib.lib()
        own = meta is None
        if own and self._img16 is not None:
            return self._img16, self._meta
        if own:
            meta = torch.zeros(2, dtype=torch.int32, device=self.device)
            _lib.check(L.tdr_screen_meta_f32(_lib.ptr(self.X), self.n, self.d, self.X.stride(0), _lib.ptr(self.norms),
                                             _lib.ptr(meta), _lib.stream_ptr()), "tdr_screen_meta_f32")
        img = torch.empty(L.tdr_packed16_floats(self.n, self.d), dtype=torch.float32, device=self.device)
        _lib.check(L.tdr_pack16_f32(_lib.ptr(self.X), self.n, self.d, self.X.stride(0), _lib.ptr(self.norms),
                                    _lib.ptr(meta), _lib.ptr(img), _lib.stream_ptr()), "tdr_pack16_f32")
        if own:
            self._img16, self._meta = img, meta
        return img, meta


PRUNE_MODE = _os.environ.get("TDR_KNN_PRUNE", "auto")  # "0": never; "force": whenever supported; "auto": N >= 65536
TILE_BOUNDS = True   # per-tile bounds as the second chance of a pruned search (ClusterIndex.tile_table); "force": always take it
_TILE_MAX_SCAN_FRACTION = 0.97
_SEED_DROP = 0.25    # adaptive seeding: a max-min SQUARED distance below a quarter of the previous one ends the seeding
_PRUNE_MIN_N = 65536
REFINE_INDEX = True            # second-chance seeds for the points far from every ball centre (ClusterIndex._refine)
_REFINE_MAX_PASSES = 3
_REFINE_MIN_SEEDS = 32         # a round that finds fewer new seeds than this is dropped (mixtures of groups of different widths: 8 seeds per
                               # round, three rounds of re-assignment for nothing: 285 -> 317 ms, tools/lab/refine_probe.py)
_REFINE_FAR = 2.0              # a point farther than this many lower-quartile radii from its centre is re-seeded (members of a ball
                               # without strays lie within ~1 such radius; in 64 dimensions the NEAREST of a few hundred foreign
                               # centres is only ~3 radii away, so the trigger ratio itself would cut nothing)
_REFINE_RADIUS_RATIO = 3.0     # ... when the largest ball radius exceeds this many lower-quartile radii
_PRUNE_MAX_SCAN_FRACTION = 0.5  # predicted share of tiles still visited above which the plain scan is used


class ClusterIndex:
    """Coarse clustering of a point block for the pruned self search: cluster-sorted row order padded to tile
    boundaries (``row_map``), one ball (centre, radius) per cluster, centre distances and the visiting order.
    Built by the package's own kernels (``csrc/tdr_cluster.hip``; nearest-centre assignments are the exact kNN kernel
    with k = 1) -- the search result does not depend on it, only the amount of work the scan can skip does."""

    _pending = None   # instances assembled from broadcast tables (row-sharded search) have nothing left to read

    @staticmethod
    def builds_at_once(N: int, n_clusters: Optional[int] = None) -> bool:
        """True when the constructor needs no host read of the seed count (the default count already equals the largest
        one the adaptive seeding may reach): ``defer=True`` then enqueues the whole build at once."""
        C = int(n_clusters or min(2048, max(8, N // 1000)))
        c_max = C if n_clusters else int(min(2048, max(C, N // 64)))
        return c_max <= C

    def __init__(self, P: "PackedPoints", n_clusters: Optional[int] = None, iters: int = 2, defer: bool = False):
        """``defer``: enqueue the build and return; ``finish()`` (the one host read, of the padded image's row count) is
        called later -- the pruned search enqueues its pilot launches in between, so that the single-workgroup seeding
        kernel is already running when the pilots fill the register files of every CU."""
        X = P.X
        dev = X.device
        N, D = X.shape
        L = _lib.lib()
        self._auto_count = n_clusters is None
        # more than 2048 balls measured slower at N = 4M (3.5 s vs 2.7 s): Gaussian blobs in high dimension are not
        # resolved further by splitting them -- the sub-balls overlap and all of them are scanned anyway
        C = int(n_clusters or min(2048, max(8, N // 1000)))
        # no count asked for: the seeding may go on past the default up to c_max seeds and stops where the max-min distance
        # collapses -- one ball per well-separated group when the data has between C and c_max of them (N = 500k in 1000
        # blobs: the default of 500 balls merges blobs in pairs, nothing can be pruned and the search is 7x slower than at
        # N = 1M, where the default happens to equal the number of blobs)
        c_max = C if n_clusters else int(min(2048, max(C, N // 64)))
        S = int(min(N, max(8 * C, min(8192, 8 * c_max)), L.tdr_cluster_maxmin_capacity()))
        C, c_max = min(C, S), min(c_max, S)
        st = _lib.stream_ptr()
        # 1-2. stratified sample; farthest-point seeds on its exact distance matrix (dense MFMA kernel + one workgroup):
        # one seed per well-separated group, an epsilon-net otherwise (random seeds leave merged clusters whose large
        # balls every workgroup would have to scan)
        sample_idx = torch.empty(S, dtype=torch.int32, device=dev)
        _lib.check(L.tdr_cluster_sample_i32(N, S, 20240917, _lib.ptr(sample_idx), st), "tdr_cluster_sample_i32")
        Xs = torch.empty((S, D), dtype=torch.float32, device=dev)
        _lib.check(L.tdr_gather_rows_f32(_lib.ptr(X), X.stride(0), D, _lib.ptr(sample_idx), None, S, _lib.ptr(Xs), st), "tdr_gather_rows_f32")
        Ps = PackedPoints(Xs)
        D2 = dense_packed(Ps, Ps, "sqeuclidean", False)
        seeds = torch.empty(c_max, dtype=torch.int32, device=dev)
        n_seeds = None
        if c_max > C:
            n_seeds = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(L.tdr_cluster_maxmin_adaptive_f32(_lib.ptr(D2), D2.stride(0), S, C, c_max, _SEED_DROP, _lib.ptr(seeds),
                                                         _lib.ptr(n_seeds), st), "tdr_cluster_maxmin_adaptive_f32")
        else:
            _lib.check(L.tdr_cluster_maxmin_f32(_lib.ptr(D2), D2.stride(0), S, C, _lib.ptr(seeds), st), "tdr_cluster_maxmin_f32")
        del D2
        # the rest of the build needs the seed count on the host when it was left to the data: with `defer` it is run by
        # finish(), AFTER the caller has enqueued its pilot searches (the read waits for the seeding kernel, the pilots
        # run meanwhile)
        self._seeded = (P, Xs, Ps, sample_idx, seeds, n_seeds, C, S, iters)
        self._seed_event = torch.cuda.Event()
        self._seed_event.record()       # finish() may run under another stream than this one
        if not (defer and n_seeds is not None):
            self._build_rest()
        if not defer:
            self.finish()

    def _build_rest(self):
        P, Xs, Ps, sample_idx, seeds, n_seeds, C, S, iters = self._seeded
        self._seeded = None
        X = P.X
        dev = X.device
        N, D = X.shape
        L = _lib.lib()
        st = _lib.stream_ptr()
        torch.cuda.current_stream(dev).wait_event(self._seed_event)
        if n_seeds is not None:
            C = int(n_seeds.item())     # the host read: the seeding kernel has finished when it returns
        cent = torch.empty((C, D), dtype=torch.float32, device=dev)
        _lib.check(L.tdr_gather_rows_f32(_lib.ptr(X), X.stride(0), D, _lib.ptr(sample_idx), _lib.ptr(seeds), C, _lib.ptr(cent), st),
                   "tdr_gather_rows_f32")
        # 3. Lloyd steps on the sample, then the assignment of all N points: nearest centre = exact kNN with k = 1
        ws = torch.empty(max(int(L.tdr_cluster_tables_workspace_bytes(N, C)) // 4, C * D + 3 * C) + 1, dtype=torch.int32, device=dev)
        for _ in range(iters):
            _, lab = knn_packed(Ps, PackedPoints(cent), 1, "sqeuclidean", exclude_self=False, _allow_screen=False)
            _lib.check(L.tdr_cluster_update_f32(_lib.ptr(Xs), S, D, _lib.ptr(lab), C, _lib.ptr(cent), _lib.ptr(ws), st),
                       "tdr_cluster_update_f32")
        self._refinable = bool(self.__dict__.get("_auto_count", False))      # the caller left the count to the data
        self._assign_and_tables(P, cent, ws)

    def _assign_and_tables(self, P, cent, ws=None):
        """Steps 3b-5 of the build for the centres `cent`: nearest-centre assignment of all points, radii, padded cluster-sorted
        layout, centre tables.  Leaves `_pending` for `finish()`."""
        X = P.X
        dev = X.device
        N, D = X.shape
        C = int(cent.shape[0])
        L = _lib.lib()
        st = _lib.stream_ptr()
        if ws is None or ws.numel() < max(int(L.tdr_cluster_tables_workspace_bytes(N, C)) // 4, C * D + 3 * C) + 1:
            ws = torch.empty(max(int(L.tdr_cluster_tables_workspace_bytes(N, C)) // 4, C * D + 3 * C) + 1, dtype=torch.int32, device=dev)
        if _opt("ASSIGN16") and L.tdr_cluster_assign16_supported(D) and N >= 65536:
            # nearest centre by the one-term screening value on the f16 matrix pipe (0.4 ms at N = 1M, C = 1000, against 3.8-6.4 ms
            # for the exact fp32 search with k = 1): the assignment shapes the clusters, no result depends on it
            x16, meta16 = P.screen_image()
            # the cached image may have been packed under another stream (the caller's): tell the allocator it is read here
            x16.record_stream(torch.cuda.current_stream(dev))
            meta16.record_stream(torch.cuda.current_stream(dev))
            c16, _ = PackedPoints(cent).screen_image(meta16)
            labels = torch.empty(N, dtype=torch.int32, device=dev)
            _lib.check(L.tdr_cluster_assign16_f32(_lib.ptr(x16), N, _lib.ptr(c16), C, D, _lib.ptr(meta16), _lib.ptr(labels), st),
                       "tdr_cluster_assign16_f32")
            labels = labels.view(N, 1)
        else:
            _, labels = knn_packed(P, PackedPoints(cent), 1, "sqeuclidean", exclude_self=False, _allow_screen=False)
        # 4-5. radii (rounded up), padded cluster-sorted layout, centre distances (rounded down), visiting order
        cap = N + 32 * C
        radius = torch.empty(C, dtype=torch.float32, device=dev)
        tile_begin = torch.empty(C + 1, dtype=torch.int32, device=dev)
        tiles = torch.empty(C, dtype=torch.int32, device=dev)
        tile_cluster = torch.empty(cap // 32, dtype=torch.int32, device=dev)
        row_map = torch.empty(cap, dtype=torch.int32, device=dev)
        n_img = torch.zeros(1, dtype=torch.int64, device=dev)
        cd = torch.empty((C, C), dtype=torch.float32, device=dev)
        order = torch.empty((C, C), dtype=torch.int32, device=dev)
        # compact cluster-sorted order (members of a cluster by ascending row: the same on every run and every rank):
        # perm[position] = row, inv[row] = position
        self.perm = torch.empty(N, dtype=torch.int32, device=dev)
        self.inv = torch.empty(N, dtype=torch.int32, device=dev)
        self.ppos = torch.empty(N, dtype=torch.int32, device=dev)   # position -> position in the padded layout
        _lib.check(
            L.tdr_cluster_tables_f32(_lib.ptr(X), N, D, X.stride(0), _lib.ptr(labels), _lib.ptr(cent), C, _lib.ptr(radius),
                                     _lib.ptr(tile_begin), _lib.ptr(tiles), _lib.ptr(tile_cluster), _lib.ptr(row_map),
                                     _lib.ptr(n_img), _lib.ptr(cd), _lib.ptr(order), _lib.ptr(self.perm), _lib.ptr(self.inv),
                                     _lib.ptr(self.ppos), _lib.ptr(ws), ws.numel() * 4, st),
            "tdr_cluster_tables_f32",
        )
        self.n_clusters = C
        self.centres = cent
        self.tile_cdist = None
        self.tile_begin = tile_begin
        self.radius = radius     # rounded up in the kernel
        self.dist = cd           # rounded down in the kernel
        self.order = order
        self.img16 = None
        self._pending = (n_img, row_map, tile_cluster, tiles)
        self._points, self._labels = P, labels

    _seeded = None

    def finish(self):
        if self._seeded is not None:
            self._build_rest()
        if self._pending is None:
            return self
        n_img, row_map, tile_cluster, tiles = self._pending
        self._pending = None
        # one host read: the padded image's row count and, when the ball count was left to the data, the largest and the median
        # radius -- a ball that absorbed a stray group carries that group's distance as its radius and cannot be pruned by anyone
        passes = int(self.__dict__.get("_refine_passes", 0))
        refinable = (bool(self.__dict__.get("_refinable", False)) and _opt("REFINE_INDEX") and self.n_clusters < 4096 - 16
                     and passes < _REFINE_MAX_PASSES)
        if refinable:
            r = self.radius
            # (the LOWER QUARTILE stands for a ball without strays: with more groups than balls most balls have absorbed one)
            r_low = r.kthvalue(max(1, int(r.numel()) // 4)).values
            n_img_v, r_max, r_med = torch.cat([n_img.double().reshape(1), r.max().double().reshape(1), r_low.double().reshape(1)]).tolist()
            LAST_KNN["index_radii"] = (round(r_max, 3), round(r_med, 3), int(r.numel()))
            n_img = int(n_img_v)
            # the re-seeding itself waits for `refine_if_needed()`: this read usually happens on the side stream of the index build,
            # beside the pilot launches, where a round's small kernels and host reads cost ~10 ms of contention (3 ms alone)
            self._refine_want = float(r_med) if (r_med > 0.0 and r_max > _REFINE_RADIUS_RATIO * r_med) else None
        else:
            n_img = int(n_img.item())
            self._refine_want = None
        if self._refine_want is None:
            self.__dict__.pop("_points", None)
            self.__dict__.pop("_labels", None)
        self.n_img = max(n_img, 32)
        self.row_map = row_map[: self.n_img]
        self.tile_cluster = tile_cluster[: self.n_img // 32]
        self._tiles_i32 = tiles
        self.tiles = tiles.to(torch.int64)
        self._scan_fraction_memo = {}
        return self

    def refine_if_needed(self):
        """Run the re-seeding rounds `finish()` found a reason for (call it where nothing else competes for the device: after the
        pilot streams have joined).  Each round ends with its own `finish()`."""
        while self.__dict__.get("_refine_want") is not None:
            r_low = self._refine_want
            self._refine_want = None
            self._refine_passes = int(self.__dict__.get("_refine_passes", 0)) + 1
            if not self._refine(r_low):
                break
            self.finish()          # reads the refined tables' size and decides about another round
        self.__dict__.pop("_points", None)
        self.__dict__.pop("_labels", None)
        return self

    def _refine(self, r_med: float) -> bool:
        """Second-chance seeds for the points the first seeding did not cover: the sample of 8192 points the seeds are chosen from
        does not contain small groups at all (3000 groups with heavy-tailed sizes: `profiles/r06_knn_regimes.jsonl`), each of them
        is absorbed by some ball and sets its radius.  Points farther than _REFINE_FAR lower-quartile radii from their centre are sampled again,
        seeded by the same farthest-point rule (up to 4096 balls in all), and everything is assigned anew.  Deterministic (the same
        index on every rank); the search result does not depend on it.  Returns False when there is nothing to refine."""
        P, labels = self._points, self._labels
        X = P.X
        dev = X.device
        N, D = X.shape
        L = _lib.lib()
        st = _lib.stream_ptr()
        cent, C = self.centres, self.n_clusters
        thr = (_REFINE_FAR * r_med) ** 2
        far = torch.empty(N, dtype=torch.bool, device=dev)
        lab = labels.reshape(-1).long()
        for b in range(0, N, 262144):      # |x - c(x)|^2 block by block (N x D temporaries only for a block)
            e = min(N, b + 262144)
            far[b:e] = ((X[b:e] - cent[lab[b:e]]) ** 2).sum(1) > thr
        res = far.nonzero().squeeze(1)
        n_res = int(res.numel())
        if n_res < max(64, N // 5000):
            return False
        def seed_far(S2, c_lo, c_hi):
            """Farthest-point seeds (adaptive count) on S2 evenly spaced far points: (sample rows, seeds, count)."""
            pick = res[torch.linspace(0, n_res - 1, S2, device=dev).long()].to(torch.int32)
            Xs2 = torch.empty((S2, D), dtype=torch.float32, device=dev)
            _lib.check(L.tdr_gather_rows_f32(_lib.ptr(X), X.stride(0), D, _lib.ptr(pick), None, S2, _lib.ptr(Xs2), st), "tdr_gather_rows_f32")
            Ps2 = PackedPoints(Xs2)
            D2 = dense_packed(Ps2, Ps2, "sqeuclidean", False)
            seeds2 = torch.empty(c_hi, dtype=torch.int32, device=dev)
            n2 = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(L.tdr_cluster_maxmin_adaptive_f32(_lib.ptr(D2), D2.stride(0), S2, c_lo, c_hi, _SEED_DROP, _lib.ptr(seeds2),
                                                         _lib.ptr(n2), st), "tdr_cluster_maxmin_adaptive_f32")
            return Xs2, seeds2, int(n2.item())

        if int(self.__dict__.get("_refine_passes", 1)) <= 1 and n_res > 1024:
            # first round: a PROBE on 1024 far points and at most 768 seeds (~2 ms) before the full round (up to 3800 dependent
            # seeding steps: 7 ms when the max-min distance never collapses, which is what far points of merely WIDE groups do)
            _, _, kp = seed_far(1024, 8, 768)      # fewer seeds than points: with as many, the rule always ends in a "collapse" (to zero)
            LAST_KNN["index_refine_probe"] = kp
            if kp < _REFINE_MIN_SEEDS:
                self._refine_passes = _REFINE_MAX_PASSES
                return False
        S2 = int(min(n_res, 8192, L.tdr_cluster_maxmin_capacity()))
        c_extra = int(min(4096 - C, S2))
        Xs2, seeds2, k2 = seed_far(S2, min(8, c_extra), c_extra)
        LAST_KNN["index_refine_try"] = (C, k2, n_res, S2)
        if k2 < _REFINE_MIN_SEEDS:
            # the far points are not separated groups: the farthest-point rule found no collapse of the max-min distance and hands back
            # its minimum count (members of WIDE groups among narrow ones, radii that differ by the data's nature; also groups whose
            # spacing shrinks gradually, as in 64 dimensions) -- nothing to gain from more balls, and no further round
            self._refine_passes = _REFINE_MAX_PASSES
            return False
        cent2 = torch.cat([cent, Xs2[seeds2[:k2].long()]]).contiguous()
        LAST_KNN["index_refined"] = (LAST_KNN.get("index_refined") or []) + [(C, C + k2, n_res)]
        self._assign_and_tables(P, cent2)
        return True

    def record_stream(self, stream):
        """The tables may have been allocated while a side stream was current (the build runs next to the pilots); the
        search that follows uses them on `stream`: tell the caching allocator, so that freeing them cannot hand the
        memory to a later side-stream allocation while kernels of `stream` still read it."""
        for t in vars(self).values():
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)
        return self

    def scan_fraction(self, tau: float) -> float:
        """Share of the database tiles a query block still has to visit when its thresholds are <= tau (squared
        distance units): clusters c with max(0, |c_w - c_c| - R_w - R_c)^2 <= tau, averaged over the blocks."""
        memo = self.__dict__.setdefault("_scan_fraction_memo", {})
        key = float(tau)
        if key in memo:      # the tier choice and the pruning decision ask for the same thresholds
            return memo[key]
        t32 = getattr(self, "_tiles_i32", None)
        if t32 is not None and self.dist.is_cuda and self.dist.dtype == torch.float32:
            # one launch, exact integer sums, one host read (tdr_cluster_scan_fraction_f32)
            out = torch.zeros(2, dtype=torch.int64, device=self.dist.device)
            _lib.check(_lib.lib().tdr_cluster_scan_fraction_f32(_lib.ptr(self.dist), _lib.ptr(self.radius), _lib.ptr(t32), self.n_clusters,
                                                                key, _lib.ptr(out), _lib.stream_ptr()), "tdr_cluster_scan_fraction_f32")
            num, tot = out.tolist()
            memo[key] = float(num) / float(tot * tot) if tot else 1.0
            return memo[key]
        gap = (self.dist - self.radius[:, None] - self.radius[None, :]).clamp_(min=0)
        t = self.tiles.to(gap.dtype)
        visited = torch.mv((gap * gap <= tau).to(gap.dtype), t)
        memo[key] = float((visited * t).sum() / (t.sum() ** 2))
        return memo[key]


    def tiles_hopeless(self, tau: float) -> bool:
        """True when no per-tile bound is expected to skip anything at threshold tau, so that the table (6-9 ms at N = 1M) is not
        built for nothing.  A tile's bound to cluster c is min over its rows of |x - c_c| - R_c; in high dimension |x - c_c|^2 ~
        |c_w - c_c|^2 + |x - c_w|^2 <= |c_w - c_c|^2 + R_w^2, so sqrt(|c_w - c_c|^2 + R_w^2) - R_c is what the BEST tile of
        cluster w can show against cluster c.  If that stays below sqrt(tau) for every pair -- one Gaussian, uniform data: centres a
        few units apart under radii and neighbour distances several times that -- the search goes straight to the plain scan ("every pair" weighted by tile counts: more
        than _TILE_MAX_SCAN_FRACTION of the tiles would still be visited).  A heuristic in one direction only: a wrong "hopeless" costs the pruning of a search that would have been pruned a little,
        never a result (blobs at centre distance 16, radius 6.6, k-th neighbour at 8: 10.7 against 8, the table is built)."""
        best = (self.dist * self.dist + (self.radius * self.radius)[:, None]).sqrt() - self.radius[None, :]
        best.fill_diagonal_(0.0)
        # share of the tiles still visited if every tile showed that best bound (weighted by the clusters' tile counts, as
        # scan_fraction does: a few small outlier clusters that could be skipped do not pay for the table)
        t = self.tiles.to(best.dtype)
        visited = torch.mv((best.clamp(min=0) ** 2 <= tau).to(best.dtype), t)
        return float((visited * t).sum() / (t.sum() ** 2)) > _TILE_MAX_SCAN_FRACTION

    # ---- per-tile bounds: the second chance of data whose balls overlap ---------------------------------------------------
    def tile_table(self, P: "PackedPoints"):
        """(n_img / 32, C) lower bounds of the distance from any row of a tile of the sorted order to every centre
        (``tdr_cluster_tile_cdist_f32`` on blocks of the exact distance matrix rows x centres from the dense MFMA kernel).  With it
        the scan skips a cluster when |x - c| - R_c of the workgroup's OWN rows exceeds their thresholds -- no query-side radius,
        and in high dimension |x - c| ~ sqrt(|c_w - c|^2 + |x - c_w|^2): blobs whose balls overlap (centre distance 16, radius
        6.6, k-th neighbour at 8: the ball-to-ball bound keeps every cluster) are still told apart (the tile bound keeps 11 %)."""
        if self.tile_cdist is not None:
            return self.tile_cdist
        self.finish()
        L = _lib.lib()
        dev, d, C = P.device, P.d, self.n_clusters
        PC = PackedPoints(self.centres)
        T = torch.empty((self.n_img // 32, C), dtype=torch.float32, device=dev)
        chunk = 32768
        for r0 in range(0, self.n_img, chunk):
            rm = self.row_map[r0:r0 + chunk]
            Pq = PackedPoints(P.X.index_select(0, rm.clamp(min=0).long()))
            D2 = dense_packed(Pq, PC, "sqeuclidean", False)
            _lib.check(L.tdr_cluster_tile_cdist_f32(_lib.ptr(D2), D2.stride(0), rm.numel(), C, d, _lib.ptr(rm), _lib.ptr(Pq.norms),
                                                    _lib.ptr(PC.norms), _lib.ptr(T[r0 // 32:]), _lib.stream_ptr()),
                       "tdr_cluster_tile_cdist_f32")
        self.tile_cdist = T
        return T

    def scan_fraction_tiles(self, tau: float) -> float:
        """Predicted share of the database tiles a query tile still visits under the per-tile bound at threshold tau (squared
        units), from ~512 tiles spread over the sorted order."""
        memo = self.__dict__.setdefault("_tiles_share_memo", {})
        if tau in memo:
            return memo[tau]
        T = self.tile_cdist
        step = max(T.shape[0] // 512, 1)
        sub = T[::step]
        sub = sub[torch.isfinite(sub[:, 0])]
        gap = (sub - self.radius[None, :]).clamp_(min=0)
        t = self.tiles.to(gap.dtype)
        visited = torch.mv((gap * gap <= tau).to(gap.dtype), t)
        memo[tau] = float(visited.mean() / t.sum())
        return memo[tau]


def _use_screen(Q, Y, nq, k, metric):
    if _opt("SCREEN_MODE") == "0" or metric not in ("sqeuclidean", "euclidean"):
        return False
    if not _lib.lib().tdr_knn_screen_supported(Y.d, k):
        return False
    if _opt("SCREEN_MODE") == "force":
        return True
    return nq * Y.n >= _SCREEN_MIN_PAIRS and Y.n >= 4096


_TIER_NAMES = ("screen-1term", "screen", "screen-long")


def _screen_operands(Q, Y):
    """fp16-split images of the query and database blocks with ONE power-of-two scale (max |x| over both, max norm over
    the database).  Returns (q16, y16, meta)."""
    L = _lib.lib()
    if Q is Y:
        y16, meta = Y.screen_image()
        return y16, y16, meta
    dev, d = Y.device, Y.d
    meta = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.check(L.tdr_screen_meta_f32(_lib.ptr(Y.X), Y.n, d, Y.X.stride(0), _lib.ptr(Y.norms), _lib.ptr(meta),
                                     _lib.stream_ptr()), "tdr_screen_meta_f32")
    _lib.check(L.tdr_screen_meta_f32(_lib.ptr(Q.X), Q.n, d, Q.X.stride(0), None, _lib.ptr(meta),
                                     _lib.stream_ptr()), "tdr_screen_meta_f32")
    y16, _ = Y.screen_image(meta)
    q16, _ = Q.screen_image(meta)
    return q16, y16, meta


def _screen_launch(Q, Y, ops, q0, nq, k, metric, exclude_self, q_offset, tier, predict_unsplit, out_d, out_i, profile):
    """One launch of the (unpruned) two-stage search on queries Q[q0:q0+nq].  Returns (flags, n_flagged) on the device."""
    L = _lib.lib()
    dev, d = Y.device, Y.d
    q16, y16, meta = ops
    t16 = L.tdr_packed16_floats(32, d)
    ws_bytes = L.tdr_knn_screen_workspace_bytes(nq, Y.n, d, k, tier)
    ws = torch.empty(max(ws_bytes, 8) // 8, dtype=torch.int64, device=dev)
    flags = torch.empty(nq, dtype=torch.int32, device=dev)
    n_flagged = torch.zeros(1, dtype=torch.int32, device=dev)
    if profile:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(
        L.tdr_knn_screen_f32(
            _lib.ptr(q16[(q0 // 32) * t16:]), _lib.ptr(Q.X[q0:]), Q.X.stride(0), _lib.ptr(Q.norms[q0:]), nq, q_offset + q0,
            _lib.ptr(y16), _lib.ptr(Y.X), Y.X.stride(0), _lib.ptr(Y.norms), Y.n, d, k, _METRIC_ID[metric],
            1 if exclude_self else 0, tier, 1 if predict_unsplit else 0, _lib.ptr(meta), _lib.ptr(out_d), _lib.ptr(out_i),
            _lib.ptr(flags), _lib.ptr(n_flagged), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(),
        ),
        "tdr_knn_screen_f32",
    )
    if profile:
        ev1.record()
        PROFILE.append((ev0, ev1, nq, _TIER_NAMES[tier]))
    return flags, n_flagged


# FLAT_SCAN: an UNPRUNED two-stage search of a large database runs as seed -> threshold passes with a select after each -> rescoring
# (csrc/tdr_knn_flat.hip, tdr_knn_screen_flat_f32) instead of the list-keeping kernel: same results bit for bit, lists of
# _FLAT_L entries per query in HBM (the one-term tier then serves data whose error band holds up to ~100 candidates).
FLAT_SCAN = True
FLAT_TWO_TERMS = True      # the two-term tier of the threshold scan between one and three terms
FLAT_FORCE_TERMS = 0       # 1 / 2 / 3: the threshold scan with that many terms whatever the pilots say (tests, measurement)
_FLAT_L = 128
FLAT_WS_LIMIT = 0          # bytes of threshold-scan workspace beyond which the queries go in blocks (0: a quarter of the free memory, >= 4 GB)
_FLAT_QUERY_BLOCK = 262144  # rows per block then (a multiple of the 32-row tile; 1024 workgroups: the chip stays full)
_FLAT_L_SHORT = 64         # list length when the list-keeping tier 0 (lists of <= 64) passed its pilot
# ASSIGN16: the cluster index assigns the points to their nearest centres with the one-term f16 kernel (tdr_cluster_assign16_f32)
# instead of the exact fp32 search with k = 1
ASSIGN16 = True


def _flat_terms(Q, Y, ops, q0, nq, k, metric, exclude_self, q_offset, tier):
    """(terms, L): number of split terms (1, 2 or 3) and list length the threshold scan should use for this search, or (0, 0)
    when it does not serve it.  `tier` = the list-keeping tier the pilot chose (-1: none passed).  Tier 0 passed: one term with
    lists of _FLAT_L_SHORT entries (the list-keeping kernel's own tier-0 lists are no longer: the band fits them, and selects and
    rescoring cost half of what lists of 128 do).  Else the one-term form (h.h') with lists of _FLAT_L when a pilot slice predicts
    that <= 5 % of the queries hold more than _FLAT_L candidates in its (wider) band; else the two-term form (h.h' + h.l': a third
    less matrix work than three terms, half of one term's band) on the same prediction made for ITS band on a three-term pilot's
    near-exact screening values; else three terms."""
    L = _lib.lib()
    if not _opt("FLAT_SCAN") or nq < _SCREEN_PILOT_MIN_Q or k > _FLAT_L - 8:
        return 0, 0
    LL = _FLAT_L
    ok = [False] + [L.tdr_knn_screen_flat_workspace_bytes(nq, Y.n, Y.d, k, t, LL) != 0 for t in (1, 2, 3)]
    forced = int(_opt("FLAT_FORCE_TERMS"))
    if forced:
        if forced not in (1, 2, 3):
            raise ValueError(f"[torchdr_amd] FLAT_FORCE_TERMS must be 0 (pilots decide), 1, 2 or 3 (got {forced}).")
        return (forced, LL) if ok[forced] else (0, 0)
    if tier == 0 and ok[1]:
        return 1, (_FLAT_L_SHORT if k + 16 <= _FLAT_L_SHORT else LL)
    if ok[1] and _flat_pilot(Q, Y, ops, q0, k, metric, exclude_self, q_offset, 0, LL) <= _SCREEN_PILOT_MAX_FRAC:
        return 1, LL
    if ok[2] and _opt("FLAT_TWO_TERMS") and _flat_pilot(Q, Y, ops, q0, k, metric, exclude_self, q_offset, 1, LL, 2) <= _SCREEN_PILOT_MAX_FRAC:
        return 2, LL
    if tier in (1, 2) and ok[3]:
        return 3, LL
    if ok[3] and _flat_pilot(Q, Y, ops, q0, k, metric, exclude_self, q_offset, 1, LL) <= _SCREEN_PILOT_MAX_FRAC:
        return 3, LL
    return 0, 0


def _flat_pilot(Q, Y, ops, q0, k, metric, exclude_self, q_offset, tier, pred_L, pred_terms=0):
    """Share of a pilot slice of queries whose error band (of list-keeping tier `tier`'s split; pred_terms = 2: of the two-term
    split, counted on tier 1's screening values) holds >= pred_L candidates."""
    L = _lib.lib()
    dev, d = Y.device, Y.d
    q16, y16, meta = ops
    nq = _SCREEN_PILOT_Q
    t16 = L.tdr_packed16_floats(32, d)
    ws_bytes = L.tdr_knn_screen_workspace_bytes(nq, Y.n, d, k, tier)
    if ws_bytes == 0:
        return 1.0
    ws = torch.empty(max(ws_bytes, 8) // 8, dtype=torch.int64, device=dev)
    pd = torch.empty((nq, k), dtype=torch.float32, device=dev)
    pi = torch.empty((nq, k), dtype=torch.int32, device=dev)
    flags = torch.empty(nq, dtype=torch.int32, device=dev)
    n_flagged = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(
        L.tdr_knn_screen_pilot_f32(
            _lib.ptr(q16[(q0 // 32) * t16:]), _lib.ptr(Q.X[q0:]), Q.X.stride(0), _lib.ptr(Q.norms[q0:]), nq, q_offset + q0,
            _lib.ptr(y16), _lib.ptr(Y.X), Y.X.stride(0), _lib.ptr(Y.norms), Y.n, d, k, _METRIC_ID[metric],
            1 if exclude_self else 0, tier, int(pred_L), int(pred_terms), _lib.ptr(meta), _lib.ptr(pd), _lib.ptr(pi),
            _lib.ptr(flags), _lib.ptr(n_flagged), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(),
        ),
        "tdr_knn_screen_pilot_f32",
    )
    return int(n_flagged.item()) / float(nq)


def _flat_launch(Q, Y, ops, q0, nq, k, metric, exclude_self, q_offset, terms, LL, out_d, out_i, profile):
    """The unpruned two-stage search as a threshold scan (tdr_knn_screen_flat_f32).  Returns (flags, n_flagged)."""
    L = _lib.lib()
    dev, d = Y.device, Y.d
    q16, y16, meta = ops
    t16 = L.tdr_packed16_floats(32, d)
    # lists and survivor regions are per query (nq * (8 L + 2060) bytes: 3.1 GB at 1M queries, 12 GB at 4M): one piece while it
    # is a small part of the free memory, else query blocks of FLAT_QUERY_BLOCK rows (independent searches, same results)
    limit = int(_opt("FLAT_WS_LIMIT")) or max(4 << 30, torch.cuda.mem_get_info(dev)[0] // 4)
    blk = nq
    if L.tdr_knn_screen_flat_workspace_bytes(nq, Y.n, d, k, terms, int(LL)) > limit and nq > _FLAT_QUERY_BLOCK:
        blk = _FLAT_QUERY_BLOCK
    ws_bytes = L.tdr_knn_screen_flat_workspace_bytes(min(blk, nq), Y.n, d, k, terms, int(LL))
    ws = torch.empty(max(ws_bytes, 8) // 8, dtype=torch.int64, device=dev)
    flags = torch.empty(nq, dtype=torch.int32, device=dev)
    n_flagged = torch.zeros(1, dtype=torch.int32, device=dev)
    if profile:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    for o in range(0, nq, blk):
        m = min(blk, nq - o)
        nf = n_flagged if blk == nq else torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(
            L.tdr_knn_screen_flat_f32(
                _lib.ptr(q16[((q0 + o) // 32) * t16:]), _lib.ptr(Q.X[q0 + o:]), Q.X.stride(0), _lib.ptr(Q.norms[q0 + o:]), m,
                q_offset + q0 + o, _lib.ptr(y16), _lib.ptr(Y.X), Y.X.stride(0), _lib.ptr(Y.norms), Y.n, d, k, _METRIC_ID[metric],
                1 if exclude_self else 0, int(terms), int(LL), _lib.ptr(meta), _lib.ptr(out_d[o:]), _lib.ptr(out_i[o:]),
                _lib.ptr(flags[o:]), _lib.ptr(nf), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(),
            ),
            "tdr_knn_screen_flat_f32",
        )
        if nf is not n_flagged:
            n_flagged += nf
    if profile:
        ev1.record()
        PROFILE.append((ev0, ev1, nq, "screen-flat%d" % terms))
    return flags, n_flagged


_SIDE_STREAMS = {}


def _side_streams(dev, n):
    """Cached side streams of a device (pilot searches and the cluster-index build run next to each other)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    have = _SIDE_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=dev))
    return have[:n]


def _pilot_tau(kth2: torch.Tensor) -> float:
    """The k-th neighbour distance (squared) the PREDICTIONS of a search are made for -- scan shares, tile bounds, list form; no
    result depends on it: the largest of the pilot rows, unless that one is an outlier (more than four times the median: rows of groups
    with fewer than k members, whose neighbours sit in other groups -- `profiles/r06_knn_regimes.jsonl`, heavy-tailed group sizes: 2 %
    of the rows made every prediction read "nothing can be pruned"); then the 90th percentile.  One host read."""
    v = kth2.float()
    mx, med, q90 = torch.stack([v.max(), v.median(), torch.quantile(v, 0.9)]).tolist()
    if med > 0.0 and mx > 4.0 * med:
        LAST_KNN["pilot_tau"] = ("q90", mx, q90)
        return float(max(q90, med))
    LAST_KNN["pilot_tau"] = ("max", mx, q90)
    return float(mx)


def _choose_tier(Q, Y, ops, q0, k, metric, exclude_self, q_offset, side_work=None, scan_frac_of=None):
    """Pilot: screen a slice of _SCREEN_PILOT_Q (512) queries with the tiers (one-term, three-term, three-term with long lists), flagging what
    an UNSLICED launch would flag; among the tiers with <= 5 % flagged the one with the smallest estimated total time wins
    (`_pick_tier`; ``scan_frac_of``: callable tau -> predicted scan share of a pruned search, evaluated after the side work
    -- the cluster index -- is complete; the long-list tier gets its pilot only when nothing passed or when the best candidate's
    predicted exact re-search of flagged rows is worth more than a pilot launch).  Returns (tier, tau) with tau the largest
    k-th neighbour distance (squared) of the slice, or (-1, None) when the worst-case band swallows the spare list slots
    for a sizeable share of the queries under every tier (large ||x|| ||y|| relative to the neighbour spacing): the
    one-stage kernel serves such data.
    A pilot launch is 256 workgroups (one per CU) of ~5 ms, so the first two tiers run CONCURRENTLY on two streams, and
    ``side_work`` (the cluster-index build, a chain of small kernels plus one single-workgroup seeding kernel) on a
    third; all are joined before the one host read of the flag counters."""
    L = _lib.lib()
    dev, d = Y.device, Y.d
    tiers = [t for t in (0, 1, 2) if L.tdr_knn_screen_workspace_bytes(_SCREEN_PILOT_Q, Y.n, d, k, t) != 0]
    main = torch.cuda.current_stream(dev)
    side = _side_streams(dev, 2)
    for sd in side:
        sd.wait_stream(main)

    def launch(tier):
        pd = torch.empty((_SCREEN_PILOT_Q, k), dtype=torch.float32, device=dev)
        pi = torch.empty((_SCREEN_PILOT_Q, k), dtype=torch.int32, device=dev)
        _, n_flagged = _screen_launch(Q, Y, ops, q0, _SCREEN_PILOT_Q, k, metric, exclude_self, q_offset, tier, True, pd, pi,
                                      profile=False)
        return pd, n_flagged

    runs = {}
    side_start, side_finish = side_work if side_work is not None else (None, None)
    if side_start is not None and _opt("PILOT_CONCURRENT"):
        # first in: its single-workgroup seeding kernel (4 ms of dependent steps) must hold a CU before the pilots'
        # workgroups take every register file (two pilots = 2 x 256 registers per lane on every SIMD); enqueued after
        # them, the whole chain waited for the pilots to drain (kernel trace at N = 1M: the main scan starts 17.4 ms into the
        # fit instead of 19.8 ms; pilots, chain and pilot rescoring now share the device and end together)
        with torch.cuda.stream(side[1]):
            side_start()
    if len(tiers) >= 2 and _opt("PILOT_CONCURRENT"):
        with torch.cuda.stream(side[0]):
            runs[tiers[0]] = launch(tiers[0])
        runs[tiers[1]] = launch(tiers[1])
    elif tiers:
        runs[tiers[0]] = launch(tiers[0])
    if side_finish is not None:
        if _opt("PILOT_CONCURRENT"):
            with torch.cuda.stream(side[1]):
                side_finish()
        else:
            side_start()
            side_finish()
    for sd in side:
        main.wait_stream(sd)
    for pd, n_flagged in runs.values():   # allocated under a side stream, read on the main one
        pd.record_stream(main)
        n_flagged.record_stream(main)
    ci_built = getattr(Y, "_cluster_index", None)
    if ci_built is not None:
        ci_built.record_stream(main)
    # every tier that keeps the flagged share <= 5 %: (tier, flagged share, tau).  The first two pilots have both run
    # already; the long-list tier is only tried when neither passes.
    cands = []
    for tier in tiers:
        if tier not in runs:
            if cands:
                break
            runs[tier] = launch(tier)
        pd, n_flagged = runs[tier]
        f = int(n_flagged.item()) / float(_SCREEN_PILOT_Q)
        if f <= _SCREEN_PILOT_MAX_FRAC:
            kth = pd[:, -1]
            cands.append((tier, f, _pilot_tau(kth * kth if metric == "euclidean" else kth)))
    nq_all = Q.n if Q is not Y else Y.n
    if cands and 2 in tiers and 2 not in runs:
        # A tier that passes with a few per cent flagged can still be the wrong one: every flagged row is recomputed by a full
        # one-stage scan (N = 1M, D = 256: 4 us per row -- 5 % flagged = 200 ms against a pruned scan of 30-70).  When the best
        # candidate's predicted re-search costs more than a pilot launch, the long-list tier gets its pilot too and competes
        # (N = 1M, D = 256, centre scale 5, k = 15: tier 1 passed at < 5 % in a 512-query pilot, flagged 6.4 % of the rows and took
        # 322 ms; tier 2 takes 69: profiles/r05_knn_pruned_matrix.jsonl)
        _, f_best, _ = _pick_tier(cands, nq_all, Y.n, d, scan_frac_of)
        if f_best * float(nq_all) * float(Y.n) * d * 2.0 / _EXACT_RATE > _LONG_TIER_PILOT_SEC:
            pd, n_flagged = runs[2] = launch(2)
            f = int(n_flagged.item()) / float(_SCREEN_PILOT_Q)
            if f <= _SCREEN_PILOT_MAX_FRAC:
                kth = pd[:, -1]
                cands.append((2, f, float((kth * kth if metric == "euclidean" else kth).max())))
    LAST_KNN["tier_candidates"] = [(t, round(f, 4)) for t, f, _ in cands]
    if not cands:
        return -1, None
    tier, _, tau = _pick_tier(cands, nq_all, Y.n, d, scan_frac_of)
    return tier, tau


# matrix work of a tier relative to the one-term tier, and the two rates of the cost model (effective flop/s of the
# one-term screening scan and of the one-stage exact kernel at the headline size)
_TIER_REL_COST = {0: 1.0, 1: 1.9, 2: 2.3}
_LONG_TIER_PILOT_SEC = 0.010     # predicted exact re-search beyond which the long-list tier's pilot (~6 ms) is worth running
_SCREEN_RATE, _EXACT_RATE = 6.2e14, 1.26e14


def _pick_tier(cands, nq, n_db, d, scan_frac_of=None):
    """Cheapest tier by estimated time = scan + exact re-search of the rows it will flag.  A flagged row costs a full
    one-stage scan of the database (n_db * d * 2 flop at the fp32 matrix rate): at N = 1M every per cent of flagged rows is
    20 ms -- next to nothing against an unpruned scan (0.4-0.8 s), but several times a PRUNED scan (15-40 ms), where the
    cheapest-tier-that-passes rule of round 2 picked the one-term tier for k = 15 and paid 50 ms for its 2 % of flagged rows
    (kNN build 84 ms; 30 ms with the three-term tier).  ``scan_frac_of(tau)``: predicted share of the tiles a pruned scan
    still visits, or None when the scan will not be pruned."""
    best = None
    for tier, f, tau in cands:
        pairs = float(nq) * float(n_db) * d * 2.0
        frac = 1.0
        if scan_frac_of is not None:
            frac = min(1.0, max(float(scan_frac_of(2.0 * tau)), 0.02))     # a pruned scan never costs less than its fixed part
        cost = pairs * _TIER_REL_COST[tier] * frac / _SCREEN_RATE + f * pairs / _EXACT_RATE
        if best is None or cost < best[0]:
            best = (cost, tier, f, tau)
    return best[1], best[2], best[3]


def _screen_fallback(Q, Y, q0, k, metric, exclude_self, q_offset, rows, out_d, out_i):
    """Recompute the flagged query rows (indices relative to q0) with the one-stage exact kernel: top-(k+1) without
    exclusion, then drop the query's own row (or the last entry when it is absent)."""
    L = _lib.lib()
    dev, d = Y.device, Y.d
    Qf = PackedPoints(Q.X[q0:][rows].contiguous())
    kk = k + 1 if exclude_self else k
    if kk > min(L.tdr_knn_max_k(d), Y.n):
        raise NotImplementedError(f"[torchdr_amd] k={k}: screening overflow fallback exceeds the exact kernel's k limit.")
    Cf, If = knn_packed(Qf, Y, kk, metric, exclude_self=False, _allow_screen=False)
    if exclude_self:
        own = (rows + (q_offset + q0)).to(torch.int32)
        is_self = If == own[:, None]
        drop = torch.where(is_self.any(1), is_self.int().argmax(1), torch.full_like(own, k, dtype=torch.int64))
        keep = torch.arange(kk, device=dev)[None, :] != drop[:, None]
        Cf = Cf[keep].view(-1, k)
        If = If[keep].view(-1, k)
    out_d[rows] = Cf
    out_i[rows] = If


def _pruned_share(Y, tau):
    """Predicted share of the tiles a pruned scan of Y still visits at threshold tau (1.0 when the index says pruning will
    not be used: the plain scan then runs)."""
    ci = getattr(Y, "_cluster_index", None)
    if ci is None:
        return 1.0
    ci.finish()
    ci.refine_if_needed()       # (the pilot streams have joined by now)
    share = ci.scan_fraction(tau)
    return share if (_opt("PRUNE_MODE") == "force" or share <= _PRUNE_MAX_SCAN_FRACTION) else 1.0


def _cluster_index_early(Y):
    """Enqueue the first stage of the index build (sample, its distance matrix, the single-workgroup seeding kernel) on the
    side stream NOW -- before the caller packs the screening image and launches its pilots.  The two pilot launches take every
    register file of the chip for ~5 ms; a seeding kernel enqueued with them waited for a CU (kernel trace: its tiny
    pack kernel sat 4.9 ms in the queue and the build ended 1.1 ms later than it had before the build was split in two)."""
    if not _opt("PILOT_CONCURRENT") or getattr(Y, "_cluster_index", None) is not None:
        return
    dev = Y.device
    side = _side_streams(dev, 2)[1]
    if ClusterIndex.builds_at_once(Y.n) and _opt("ASSIGN16"):
        # N >= 2 048 000: the seed count is fixed, so the constructor runs the WHOLE build here and now, on the side stream --
        # including the f16 nearest-centre assignment, which reads the block's screening image.  That image is cached on the
        # block and shared with the pilots (main stream, first side stream): pack it on the main stream FIRST, so that every
        # later reader is ordered behind the pack by the wait below / the waits of `_choose_tier` (packed by the
        # side stream, the pilots read a half-written image and chose tier and tau from garbage)
        Y.screen_image()
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        _cluster_index_start(Y)


def _cluster_index_start(Y):
    """Enqueue the index build (no host read yet); `_cluster_index` completes it."""
    ci = getattr(Y, "_cluster_index", None)
    if ci is None:
        ci = Y._cluster_index = ClusterIndex(Y, defer=True)
    return ci


def _cluster_index(Y, ops, build=True, refine=True):
    ci = getattr(Y, "_cluster_index", None)
    if ci is None and build:
        ci = Y._cluster_index = ClusterIndex(Y)
    if ci is not None:
        ci.finish()
        if refine:
            ci.refine_if_needed()
        ci.record_stream(torch.cuda.current_stream(Y.device))
    if ci is not None and ci.img16 is None:
        L = _lib.lib()
        ci.img16 = torch.empty(L.tdr_packed16_floats(ci.n_img, Y.d), dtype=torch.float32, device=Y.device)
        _lib.check(L.tdr_pack16_mapped_f32(_lib.ptr(Y.X), ci.n_img, Y.d, Y.X.stride(0), _lib.ptr(Y.norms), _lib.ptr(ops[2]),
                                           _lib.ptr(ci.row_map), _lib.ptr(ci.img16), _lib.stream_ptr()),
                   "tdr_pack16_mapped_f32")
    return ci


PRUNED_LISTS = "auto"       # candidate lists of the pruned scan: "lazy" buffers, "sorted" lists, "auto" = by the predicted scan share
_LAZY_MAX_SHARE = 0.05      # predicted share of the tiles still visited up to which the lazy buffers are taken: at N = 1M they were
                            # 1.4-3x faster on every search predicted at <= 0.028 and 0.55x at 0.23 / 0.49 (profiles/r06_knn_lists_matrix.jsonl)


def _pruned_launch(Y, ops, ci, k, metric, exclude_self, tier, out_d, out_i, pos_range=(0, 0), tile_cdist=None, share=None):
    """Cluster-pruned self search of Y (all of it, or the queries at positions pos_range of the sorted order): rows of
    out_d / out_i are indexed by SOURCE row.  Returns (flags indexed by source row, n_flagged).
    ``share``: the predicted share of the database tiles a workgroup still visits (None: unknown).  The lazy candidate buffers
    (one workgroup per CU, 125 entries per query) win where a query's work is its own cluster -- list maintenance -- and lose
    where most of it is the steady scan of many clusters with few survivors, which the sorted-list kernel runs with two
    workgroups per CU (`profiles/r06_knn_lists_matrix.jsonl`)."""
    L = _lib.lib()
    dev, d = Y.device, Y.d
    want = _opt("PRUNED_LISTS")
    lazy = want == "lazy" or (want != "sorted" and (share is None or share <= _LAZY_MAX_SHARE))
    if lazy and want != "lazy" and (LAST_KNN.get("pilot_tau") or ("max",))[0] == "q90":
        # the share was predicted for the 90th percentile of the pilot's k-th distances: a tenth of the rows have (much) farther
        # neighbours and scan steadily -- the sorted lists' regime (heavy-tailed group sizes: 249 ms lazy, 184 ms sorted)
        lazy = False
    LAST_KNN["predicted_share"], LAST_KNN["lists"] = share, "lazy" if lazy else "sorted"
    prev_lists = L.tdr_knn_screen_clustered_lists(-1)
    if not prev_lists:
        lazy = False            # the library-wide switch is off (measurement runs of the sorted lists)
    L.tdr_knn_screen_clustered_lists(1 if lazy else 0)
    try:
        return _pruned_launch_impl(L, Y, ops, ci, k, metric, exclude_self, tier, out_d, out_i, pos_range, tile_cdist)
    finally:
        L.tdr_knn_screen_clustered_lists(prev_lists)


def _pruned_launch_impl(L, Y, ops, ci, k, metric, exclude_self, tier, out_d, out_i, pos_range, tile_cdist):
    dev, d = Y.device, Y.d
    ws_bytes = L.tdr_knn_screen_clustered_workspace_bytes(ci.n_img, d, k, tier)
    ws = torch.empty(max(ws_bytes, 8) // 8, dtype=torch.int64, device=dev)
    flags = torch.zeros(Y.n, dtype=torch.int32, device=dev)
    n_flagged = torch.zeros(1, dtype=torch.int32, device=dev)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(
        L.tdr_knn_screen_clustered_tb_f32(
            _lib.ptr(ci.img16), _lib.ptr(Y.X), Y.X.stride(0), _lib.ptr(Y.norms), ci.n_img, d, k, _METRIC_ID[metric],
            1 if exclude_self else 0, tier, _lib.ptr(ops[2]), _lib.ptr(ci.row_map), ci.n_clusters, _lib.ptr(ci.tile_cluster),
            _lib.ptr(ci.tile_begin), _lib.ptr(ci.radius), _lib.ptr(ci.dist), _lib.ptr(ci.order),
            None if tile_cdist is None else _lib.ptr(tile_cdist), int(pos_range[0]),
            int(pos_range[1]), _lib.ptr(out_d), _lib.ptr(out_i), _lib.ptr(flags), _lib.ptr(n_flagged), _lib.ptr(ws),
            ws_bytes, _lib.stream_ptr(),
        ),
        "tdr_knn_screen_clustered_tb_f32",
    )
    if PROFILE is not None:
        ev1.record()
        n_q = Y.n if pos_range[1] <= pos_range[0] else int(pos_range[1] - pos_range[0])
        PROFILE.append((ev0, ev1, n_q, _TIER_NAMES[tier] + "-pruned"))
    return flags, n_flagged


def _want_prune(Y, n):
    return _opt("PRUNE_MODE") != "0" and (_opt("PRUNE_MODE") == "force" or n >= _PRUNE_MIN_N)


def _knn_screen(Q, Y, q0, nq, k, metric, exclude_self, q_offset, out_d, out_i, pilot=True, tier=1, info=None):
    """Two-stage search of queries Q[q0:q0+nq] against Y (tier chosen by a pilot slice for large searches; the full
    self search of clustered data additionally prunes by cluster bounds); rows whose screening list overflowed are
    redone by the one-stage exact kernel.  Returns the number of such rows, or -1 when the pilot says the data does not
    suit screening (nothing written; the caller uses the one-stage kernel)."""
    prune = Q is Y and q0 == 0 and q_offset == 0 and nq == Y.n and _want_prune(Y, Y.n)
    if prune and pilot and nq >= _SCREEN_PILOT_MIN_Q:
        _cluster_index_early(Y)
    ops = _screen_operands(Q, Y)
    pilot_tau = None
    tile_tab = None
    if pilot and nq >= _SCREEN_PILOT_MIN_Q:
        # the index build does not depend on the pilot's outcome: it runs next to it
        tier, pilot_tau = _choose_tier(Q, Y, ops, q0, k, metric, exclude_self, q_offset,
                                       side_work=((lambda: _cluster_index_start(Y)), (lambda: _cluster_index(Y, ops, refine=False))) if prune else None,
                                       scan_frac_of=(lambda tau: _pruned_share(Y, tau)) if prune else None)
        if tier < 0:
            # no list-keeping tier holds the band: the threshold scan keeps longer lists (in HBM) and may still serve it
            terms, flat_L = _flat_terms(Q, Y, ops, q0, nq, k, metric, exclude_self, q_offset, -1)
            if terms == 0:
                return -1
            flags, n_flagged = _flat_launch(Q, Y, ops, q0, nq, k, metric, exclude_self, q_offset, terms, flat_L, out_d, out_i,
                                            profile=PROFILE is not None)
            bad = int(n_flagged.item())
            if bad:
                _screen_fallback(Q, Y, q0, k, metric, exclude_self, q_offset, flags.nonzero().squeeze(1), out_d, out_i)
            LAST_KNN["tier"], LAST_KNN["pruned"], LAST_KNN["tile_bounds"], LAST_KNN["flat_terms"] = tier, False, False, terms
            if info is not None:
                info["cluster_order"] = None
            return bad
    if prune:
        ci = _cluster_index(Y, ops)
        # worth it only when the cluster balls are far apart relative to the neighbour distances: predicted from the
        # pilot's k-th distances (with slack for the blocks the pilot did not see)
        if _opt("PRUNE_MODE") != "force" and (pilot_tau is None or ci.scan_fraction(2.0 * pilot_tau) > _PRUNE_MAX_SCAN_FRACTION):
            # the balls overlap too much for the ball-to-ball bound.  Second chance: the per-tile table (a few ms of dense
            # distances rows x centres) and its own prediction
            prune = False
            if pilot_tau is not None and _opt("TILE_BOUNDS") and (_opt("TILE_BOUNDS") == "force" or not ci.tiles_hopeless(pilot_tau)):
                tile_tab = ci.tile_table(Y)
                # pilot_tau is the LARGEST k-th distance of the pilot rows and a workgroup prunes with its own, smaller,
                # thresholds, so the prediction is pessimistic -- measured at N = 1M, D = 128 (unpruned -> tile bounds, ms):
                # prediction 0.34: 497 -> 158; 0.70 (D = 64): 429 -> 242; 0.96 (N = 300k): 88 -> 59; 1.00 (blobs that
                # overlap entirely): 483 -> 448; 0.998 (ONE Gaussian): 460 -> 514.  Taken whenever anything is predicted to go.
                prune = _opt("TILE_BOUNDS") == "force" or ci.scan_fraction_tiles(pilot_tau) <= _TILE_MAX_SCAN_FRACTION
                if not prune:
                    tile_tab = None
    flat_terms = 0
    if prune:
        share = None
        if pilot_tau is not None:
            share = ci.scan_fraction_tiles(pilot_tau) if tile_tab is not None else ci.scan_fraction(2.0 * pilot_tau)
        flags, n_flagged = _pruned_launch(Y, ops, ci, k, metric, exclude_self, tier, out_d, out_i, tile_cdist=tile_tab, share=share)
    else:
        if pilot and nq >= _SCREEN_PILOT_MIN_Q:
            flat_terms, flat_L = _flat_terms(Q, Y, ops, q0, nq, k, metric, exclude_self, q_offset, tier)
        if flat_terms:
            flags, n_flagged = _flat_launch(Q, Y, ops, q0, nq, k, metric, exclude_self, q_offset, flat_terms, flat_L, out_d, out_i,
                                            profile=PROFILE is not None)
        else:
            flags, n_flagged = _screen_launch(Q, Y, ops, q0, nq, k, metric, exclude_self, q_offset, tier, False, out_d, out_i,
                                              profile=PROFILE is not None)
    bad = int(n_flagged.item())
    if bad:
        _screen_fallback(Q, Y, q0, k, metric, exclude_self, q_offset, flags.nonzero().squeeze(1), out_d, out_i)
    LAST_KNN["tier"] = tier
    LAST_KNN["pruned"] = bool(prune)
    LAST_KNN["tile_bounds"] = tile_tab is not None
    LAST_KNN["flat_terms"] = flat_terms
    # the cluster-sorted row order of a pruned self search (perm: position -> source row, inv: row -> position; members
    # of a cluster by ascending row): callers that go on to gather rows by neighbour index (the UMAP loop) renumber the
    # points in it.  Handed to the caller through its `info` record, not through module state.
    if info is not None:
        info["cluster_order"] = (ci.perm, ci.inv) if prune else None
    return bad


def _ivf_request(backend, self_search, k, n, d, metric):
    """(nlist, nprobe) when `backend` asks for an approximate index and the search is one the IVF kernel serves (full self
    search with k, sqeuclidean / euclidean, D <= 256, k within the screening lists), else None -> the exact search, which
    is a valid answer to any approximate request."""
    if backend is None or not getattr(backend, "index_type", "Flat") in ("IVF", "IVFPQ"):
        return None
    if not self_search or k is None or k >= n or metric not in ("sqeuclidean", "euclidean"):
        return None
    if d > 256 or not _lib.lib().tdr_knn_screen_supported(d, int(k)) or n < 4096:
        return None
    # the index builder's limits: the seeding workgroup's sample capacity and the 4096 clusters of the table kernel
    # (a larger nlist is served with 4096 lists and proportionally fewer probes would be wrong: keep nprobe as given)
    nlist = max(1, min(int(getattr(backend, "nlist", 100)), n // 64, int(_lib.lib().tdr_cluster_maxmin_capacity()), _IVF_MAX_LISTS))
    return nlist, max(1, min(int(getattr(backend, "nprobe", 1)), nlist))


_IVF_MAX_LISTS = 4096   # tdr_cluster_tables_f32 (per-cluster LDS rank sort of the centre distances)


def _knn_ivf(Y: "PackedPoints", k: int, metric: str, exclude_self: bool, nlist: int, nprobe: int, info=None):
    """Approximate self search (distance/faiss.py:331-349, ``IndexIVFFlat``): ``nlist`` clusters from the package's own
    index builder, ``nprobe`` cluster scans per block of 128 queries of the cluster-sorted order (``tdr_knn_ivf_f32``).
    Distances are exact for every returned pair.  Rows that find fewer than k candidates in the lists they probed (small
    or singleton clusters at low ``nprobe``; Faiss pads such rows with -1) are searched exactly instead: the affinity and
    symmetrisation stages downstream index the embedding with these columns and must never see -1 / +inf."""
    L = _lib.lib()
    dev, d = Y.device, Y.d
    ops = _screen_operands(Y, Y)
    tier = 1
    if Y.n >= _SCREEN_PILOT_MIN_Q:
        tier, _ = _choose_tier(Y, Y, ops, 0, k, metric, exclude_self, 0)
        if tier < 0:   # data the screening stage cannot serve: exact one-stage search
            return knn_packed(Y, Y, k, metric, exclude_self, _allow_screen=False)
    cache = Y.__dict__.setdefault("_ivf_index", {})
    ci = cache.get(nlist)
    if ci is None:
        ci = cache[nlist] = ClusterIndex(Y, n_clusters=nlist)
    if ci.img16 is None:
        ci.img16 = torch.empty(L.tdr_packed16_floats(ci.n_img, d), dtype=torch.float32, device=dev)
        _lib.check(L.tdr_pack16_mapped_f32(_lib.ptr(Y.X), ci.n_img, d, Y.X.stride(0), _lib.ptr(Y.norms), _lib.ptr(ops[2]),
                                           _lib.ptr(ci.row_map), _lib.ptr(ci.img16), _lib.stream_ptr()), "tdr_pack16_mapped_f32")
    out_d = torch.full((Y.n, k), float("inf"), dtype=torch.float32, device=dev)
    out_i = torch.full((Y.n, k), -1, dtype=torch.int32, device=dev)
    ws_bytes = L.tdr_knn_screen_clustered_workspace_bytes(ci.n_img, d, k, tier)
    ws = torch.empty(max(ws_bytes, 8) // 8, dtype=torch.int64, device=dev)
    flags = torch.zeros(Y.n, dtype=torch.int32, device=dev)
    n_flagged = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(
        L.tdr_knn_ivf_f32(_lib.ptr(ci.img16), _lib.ptr(Y.X), Y.X.stride(0), _lib.ptr(Y.norms), ci.n_img, d, k, _METRIC_ID[metric],
                          1 if exclude_self else 0, tier, _lib.ptr(ops[2]), _lib.ptr(ci.row_map), ci.n_clusters,
                          _lib.ptr(ci.tile_cluster), _lib.ptr(ci.tile_begin), _lib.ptr(ci.radius), _lib.ptr(ci.dist), _lib.ptr(ci.order),
                          int(nprobe), _lib.ptr(out_d), _lib.ptr(out_i), _lib.ptr(flags), _lib.ptr(n_flagged), _lib.ptr(ws), ws_bytes,
                          _lib.stream_ptr()),
        "tdr_knn_ivf_f32",
    )
    # spare list slots overflowed, or fewer than k candidates found: those rows are searched exactly
    redo = ((flags != 0) | (out_i[:, k - 1] < 0)).nonzero().squeeze(1)
    bad = int(redo.numel())
    if bad:
        _screen_fallback(Y, Y, 0, k, metric, exclude_self, 0, redo, out_d, out_i)
    LAST_KNN["path"], LAST_KNN["flagged"], LAST_KNN["tier"], LAST_KNN["pruned"] = f"ivf (nlist={ci.n_clusters}, nprobe={nprobe})", bad, tier, False
    if info is not None:
        info["cluster_order"] = (ci.perm, ci.inv)
    return out_d, out_i


def knn_pruned_sharded(Y: "PackedPoints", k: int, metric: str, exclude_self: bool, ctx: DistributedContext,
                       loop_order: bool = False, info: Optional[dict] = None):
    """Row-sharded self search with cluster-bound pruning.  Pruning works on the cluster-sorted order, so every rank
    answers the queries of ONE CONTIGUOUS RANGE of that order (the chunk rule applied to positions: balanced, arbitrary
    source rows).  Every rank builds the SAME cluster index by itself, next to the pilot search of its own chunk -- the
    index kernels are deterministic (stable layout, ordered Lloyd sums), so nothing is built on one rank and broadcast;
    the pilots' votes are combined by one MAX all-reduce, after which every decision is a function of identical inputs.

    ``loop_order=False``: the rows then travel to the ranks that own them in the CALLER's numbering (all-to-all-v, the
    exchange the symmetrisation uses); returns this rank's chunk (values, indices).
    ``loop_order=True``: no exchange -- the rank KEEPS its range.  Returned row j is position ``chunk_start + j`` of the
    cluster-sorted order and neighbour indices are positions too; ``info["cluster_order"] = (perm, inv)`` maps positions
    to the caller's rows and back.  The estimator runs every later stage (sigma search, symmetrisation, loop) in that
    numbering, where a rank's neighbours are almost all its own rows, and un-permutes the embedding at the end.

    Returns None when the ranks agree that pruning does not pay."""
    from torchdr_amd.parallel import allreduce_max_, exchange_rows_to_owners
    from torchdr_amd.utils.phases import phase

    dev = Y.device
    n, W, rank = Y.n, ctx.world_size, ctx.rank
    c0, c1 = ctx.compute_chunk_bounds(n)
    with phase("knn: pilots + cluster index"):
        _cluster_index_early(Y)
        ops = _screen_operands(Y, Y)
        q0 = max(0, min((c0 // 32) * 32, ((n - _SCREEN_PILOT_Q) // 32) * 32))
        tier, tau = _choose_tier(Y, Y, ops, q0, k, metric, exclude_self, 0,
                                 side_work=((lambda: _cluster_index_start(Y)), (lambda: _cluster_index(Y, ops, refine=False))),
                                 scan_frac_of=lambda t: _pruned_share(Y, t))
    # one decision for all ranks: any rank without a usable tier -> nobody prunes; else the most conservative tier and
    # the largest threshold estimate (element-wise MAX all-reduce)
    vote = torch.tensor([float(tier < 0), float(max(tier, 0)), 0.0 if tau is None else tau], dtype=torch.float64, device=dev)
    with phase("knn: vote (all-reduce)"):
        allreduce_max_(vote)
    if float(vote[0]) > 0:
        return None
    tier, tau = int(vote[1]), float(vote[2])
    ci = _cluster_index(Y, ops)
    tile_tab = None
    if _opt("PRUNE_MODE") != "force" and ci.scan_fraction(2.0 * tau) > _PRUNE_MAX_SCAN_FRACTION:
        # same tables and tau on every rank: same decision.  Second chance as in the single-process search: the per-tile
        # bounds (every rank builds the same table -- a few ms -- and reads the same prediction off it)
        if not _opt("TILE_BOUNDS"):
            return None
        tile_tab = ci.tile_table(Y)
        if not (_opt("TILE_BOUNDS") == "force" or ci.scan_fraction_tiles(tau) <= _TILE_MAX_SCAN_FRACTION):
            return None
    # this rank's positions [c0, c1) of the compact sorted order -> the range of the padded layout that holds them
    # (begin rounded down to a query batch; the few extra rows are answered twice, by this rank and by its neighbour)
    pp = ci.ppos[torch.tensor([c0, c1 - 1], device=dev)].tolist()
    p0, p1 = (pp[0] // 256) * 256, pp[1] + 1
    out_d = torch.empty((n, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((n, k), dtype=torch.int32, device=dev)
    rows = ci.perm[c0:c1].long()
    with phase("knn: pruned scan + rescoring"):
        share = ci.scan_fraction_tiles(tau) if tile_tab is not None else ci.scan_fraction(2.0 * tau)    # the same on every rank
        flags, n_flagged = _pruned_launch(Y, ops, ci, k, metric, exclude_self, tier, out_d, out_i, pos_range=(p0, p1),
                                          tile_cdist=tile_tab, share=share)
        if int(n_flagged.item()):
            mine = rows[flags[rows] != 0]
            if mine.numel():
                _screen_fallback(Y, Y, 0, k, metric, exclude_self, 0, mine, out_d, out_i)
    LAST_KNN["path"], LAST_KNN["flagged"], LAST_KNN["tier"], LAST_KNN["pruned"] = "screen-pruned", 0, tier, True
    if loop_order:
        with phase("knn: renumber"):
            Cc = out_d[rows]
            Ic = ci.inv[out_i[rows].long()]
        if info is not None:
            info["cluster_order"] = (ci.perm, ci.inv)
            info["loop_order"] = True
        return Cc, Ic
    with phase("knn: rows to owners (all-to-all)"):
        Cc, Ic = exchange_rows_to_owners(rows.to(torch.int32), out_d[rows], out_i[rows], n, W, c0, c1 - c0)
    return Cc, Ic


def knn_packed(
    Q: PackedPoints, Y: PackedPoints, k: int, metric: str, exclude_self: bool, q_offset: int = 0,
    q_rows: Optional[slice] = None, _allow_screen: bool = True, info: Optional[dict] = None,
):
    """k nearest database rows of ``Y`` for the queries ``Q`` (or the row slice ``q_rows`` of Q,
    which must start on a multiple of 32).  Returns (values (n,k) fp32 ascending, indices (n,k) int32).
    Large searches take the two-stage path (fp16-split screening + exact rescoring); its results are
    bit-identical to the one-stage kernel's."""
    L = _lib.lib()
    if Q.d != Y.d:
        raise ValueError("[TorchDR] ERROR : X and Y must have the same number of features.")
    d = Q.d
    if q_rows is None:
        q0, q1 = 0, Q.n
    else:
        q0, q1 = q_rows.start, q_rows.stop
        if q0 % 32 != 0:
            raise ValueError("[torchdr_amd] query slices must start on a multiple of 32 rows.")
    nq = q1 - q0
    kmax = L.tdr_knn_max_k(d)
    if k > kmax:
        # more neighbours than the scan kernels' LDS-resident lists hold (e.g. perplexity > 40 at D = 128): blocks of the
        # exact distance matrix from the dense MFMA kernel + the running top-k merge -- same bits as the scan kernels
        return _knn_dense_merge(Q, Y, q0, q1, k, metric, exclude_self, q_offset)
    dev = Y.device
    out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
    if _allow_screen and _use_screen(Q, Y, nq, k, metric):
        bad = _knn_screen(Q, Y, q0, nq, k, metric, exclude_self, q_offset, out_d, out_i, info=info)
        if bad >= 0:
            LAST_KNN["path"], LAST_KNN["flagged"] = ("screen-pruned" if LAST_KNN.get("pruned") else "screen"), bad
            return out_d, out_i
        LAST_KNN["path"], LAST_KNN["flagged"] = "exact (pilot overflow)", 0
    elif _allow_screen:
        LAST_KNN["path"], LAST_KNN["flagged"] = "exact", 0
    ws_bytes = L.tdr_knn_workspace_bytes(nq, Y.n, k)
    ws = torch.empty(max(ws_bytes, 8) // 8, dtype=torch.int64, device=dev)
    tile_floats = L.tdr_packed_floats(32, d)
    qdata = Q.data[(q0 // 32) * tile_floats:]
    prof = PROFILE is not None and _allow_screen   # internal searches (cluster index: k = 1 vs the centres) are not the kNN build's scan
    if prof:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(
        L.tdr_knn_packed_f32(
            _lib.ptr(qdata), nq, q_offset + q0, _lib.ptr(Y.data), Y.n, d, k, _METRIC_ID[metric],
            1 if exclude_self else 0, _lib.ptr(out_d), _lib.ptr(out_i), _lib.ptr(ws), ws_bytes,
            _lib.stream_ptr(),
        ),
        "tdr_knn_packed_f32",
    )
    if prof:
        ev1.record()
        PROFILE.append((ev0, ev1, nq, "exact"))
    return out_d, out_i


_F64_METRIC = {"sqeuclidean": 0, "euclidean": 1, "angular": 2}


def _pairwise_f64(X, Y, metric, exclude_diag, k, return_indices, device, distributed_ctx):
    """float64 inputs on the float64 kernels (``tdr_knn_f64``: fp64 matrix pipe + in-kernel top-k, or the dense matrix),
    as the reference computes in the dtype of its input.  Returns NotImplemented when no float64 kernel covers the call."""
    if metric not in _F64_METRIC or X.shape[1] > 256:
        return NotImplemented
    sharded = distributed_ctx is not None and distributed_ctx.is_initialized
    if sharded and (k is None or Y is not None):
        return NotImplemented     # the float32 path raises the reference's errors for these (base.py:160-175)
    L = _lib.lib()
    X = _to_device(X, device)
    _lib.require_gpu(X, "X")
    self_search = Y is None
    if sharded:
        # row-sharded (base.py:160-211): queries = this rank's chunk, database = all rows; the kernel excludes the
        # query's own row by its global index (q_global0)
        n_all = X.shape[0]
        c0, c1 = distributed_ctx.compute_chunk_bounds(n_all)
        kk = int(k)
        if kk > n_all - (1 if exclude_diag else 0) or L.tdr_knn_f64_lds_bytes(X.shape[1], kk) == 0:
            return NotImplemented
        Xc = X if X.stride(1) == 1 else X.contiguous()
        Q = Xc[c0:c1]
        ws = torch.empty((c1 - c0) + n_all, dtype=torch.float64, device=Xc.device)
        out_d = torch.empty((c1 - c0, kk), dtype=torch.float64, device=Xc.device)
        out_i = torch.empty((c1 - c0, kk), dtype=torch.int32, device=Xc.device)
        _lib.check(
            L.tdr_knn_f64(_lib.ptr(Q), c1 - c0, Q.stride(0), c0, _lib.ptr(Xc), n_all, Xc.stride(0), X.shape[1], kk, _F64_METRIC[metric],
                          1 if exclude_diag else 0, _DIAG_ADD, _lib.ptr(out_d), _lib.ptr(out_i), kk, _lib.ptr(ws), _lib.stream_ptr()),
            "tdr_knn_f64",
        )
        LAST_KNN["path"], LAST_KNN["flagged"] = "f64", 0
        return (out_d, out_i) if return_indices else out_d
    Yd = X if self_search else _to_device(Y, device).to(torch.float64)
    if X.shape[1] != Yd.shape[1]:
        raise ValueError("[TorchDR] ERROR : X and Y must have the same number of features.")
    Xc = X if X.stride(1) == 1 else X.contiguous()
    Yc = Xc if self_search else (Yd if Yd.stride(1) == 1 else Yd.contiguous())
    nq, d = Xc.shape
    n_db = Yc.shape[0]
    do_exclude = bool(exclude_diag) and self_search
    kk = int(k) if (k is not None and k < n_db) else 0
    if kk and do_exclude and kk > n_db - 1:
        raise ValueError("[TorchDR] ERROR : k must be smaller than the number of samples.")
    if L.tdr_knn_f64_lds_bytes(d, kk) == 0:
        return NotImplemented
    ws = torch.empty(nq + n_db, dtype=torch.float64, device=Xc.device)
    if kk:
        out_d = torch.empty((nq, kk), dtype=torch.float64, device=Xc.device)
        out_i = torch.empty((nq, kk), dtype=torch.int32, device=Xc.device)
        ldo = kk
    else:
        out_d = torch.empty((nq, n_db), dtype=torch.float64, device=Xc.device)
        out_i, ldo = None, n_db
    _lib.check(
        L.tdr_knn_f64(_lib.ptr(Xc), nq, Xc.stride(0), 0, _lib.ptr(Yc), n_db, Yc.stride(0), d, kk, _F64_METRIC[metric],
                      1 if do_exclude else 0, _DIAG_ADD, _lib.ptr(out_d), _lib.ptr(out_i), ldo, _lib.ptr(ws), _lib.stream_ptr()),
        "tdr_knn_f64",
    )
    LAST_KNN["path"], LAST_KNN["flagged"] = "f64", 0
    if return_indices:
        return out_d, out_i
    return out_d


def dense_packed(Q: PackedPoints, Y: PackedPoints, metric: str, exclude_self: bool, q_offset: int = 0):
    L = _lib.lib()
    out = torch.empty((Q.n, Y.n), dtype=torch.float32, device=Y.device)
    _lib.check(
        L.tdr_dense_dist_packed_f32(
            _lib.ptr(Q.data), Q.n, q_offset, _lib.ptr(Y.data), Y.n, Q.d, _METRIC_ID[metric],
            1 if exclude_self else 0, _DIAG_ADD, _lib.ptr(out), out.stride(0), _lib.stream_ptr(),
        ),
        "tdr_dense_dist_packed_f32",
    )
    return out


def _to_device(X, device):
    if device != "auto" and str(X.device) != str(device):
        X = X.to(device)
    return X


class WidePackedPoints:
    """Tile images of a point block with more than 256 features (``tdr_pack_rows_wide_f32``): same layout as
    ``PackedPoints`` with the feature dimension padded to a multiple of 32."""

    def __init__(self, X: torch.Tensor):
        _lib.require_gpu(X, "X")
        if X.dtype != torch.float32:
            raise NotImplementedError(f"[torchdr_amd] only float32 inputs are supported by the HIP distance kernels (got {X.dtype}).")
        if X.stride(1) != 1:
            X = X.contiguous()
        L = _lib.lib()
        self.n, self.d = X.shape
        self.data = torch.empty(L.tdr_packed_floats_wide(self.n, self.d), dtype=torch.float32, device=X.device)
        self.norms = torch.empty(self.n, dtype=torch.float32, device=X.device)
        _lib.check(L.tdr_pack_rows_wide_f32(_lib.ptr(X), self.n, self.d, X.stride(0), _lib.ptr(self.data), _lib.ptr(self.norms),
                                            _lib.stream_ptr()), "tdr_pack_rows_wide_f32")
        self.device, self.X = X.device, X


def _knn_wide(Xq, Y, k, metric, exclude_self, q_global0=0):
    """Exact kNN for D > 256 (sqeuclidean / euclidean / angular) on the K-chunked MFMA scan (``tdr_knn_wide_f32``):
    the reference's op sequence (distance/torch.py:82-120 + utils/utils.py:215) with the running top-k fused -- no
    block of X Y^T is ever written.  Every distance is one k-ordered fp32 fma chain over the row, which is what MKL
    computes for K <= ~380; beyond that MKL splits the contraction, so parity with the CPU reference is to fp32
    rounding, not bit for bit.  Returns None when k exceeds the kernel's LDS-resident lists."""
    L = _lib.lib()
    if k > min(int(L.tdr_knn_wide_max_k()), Y.shape[0]):
        return None
    Yp = WidePackedPoints(Y)
    Qp = Yp if Xq is Y else WidePackedPoints(Xq)
    nq, nd, d = Qp.n, Yp.n, Yp.d
    out_d = torch.empty((nq, k), dtype=torch.float32, device=Yp.device)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=Yp.device)
    ws_bytes = int(L.tdr_knn_workspace_bytes(nq, nd, k))
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=Yp.device)
    _lib.check(
        L.tdr_knn_wide_f32(_lib.ptr(Qp.data), nq, q_global0, _lib.ptr(Yp.data), nd, d, k, _METRIC_ID[metric],
                           1 if exclude_self else 0, _lib.ptr(out_d), _lib.ptr(out_i), _lib.ptr(ws), ws_bytes, _lib.stream_ptr()),
        "tdr_knn_wide_f32",
    )
    LAST_KNN["path"], LAST_KNN["flagged"] = "wide (K-chunked MFMA scan)", 0
    return out_d, out_i


def _knn_dense_merge(Q: "PackedPoints", Y: "PackedPoints", q0: int, q1: int, k: int, metric: str, exclude_self: bool, q_offset: int):
    """Exact kNN with k beyond the scan kernels' lists (``TSNE(perplexity=100)`` asks for k = 300,
    affinity/entropic.py:259; utils/utils.py:203-216 has no limit): per (4096 queries x 65536 rows) block the dense
    fp32-MFMA kernel (``tdr_dense_dist_packed_f32``: the k-ordered fma chain and the reference's epilogue, the arithmetic
    of the scan kernels and of the CPU oracle) writes the distances and ``tdr_topk_merge_f32`` (value passthrough) folds
    them into the running lists in the canonical (distance, index) order.  Distances AND indices are bit-identical to the
    scan kernels' and the oracle's -- the library-GEMM form this replaces agreed to fp32 rounding only.  k <= 1024."""
    L = _lib.lib()
    dev, d = Y.device, Y.d
    nq, nd = q1 - q0, Y.n
    if k > int(L.tdr_topk_max_k()):
        return _knn_dense_wide(Q, Y, q0, q1, k, metric, exclude_self, q_offset)
    st = _lib.stream_ptr()
    keys = torch.empty((nq, k), dtype=torch.int64, device=dev)
    _lib.check(L.tdr_topk_init(_lib.ptr(keys), nq, k, st), "tdr_topk_init")
    tile = int(L.tdr_packed_floats(32, d))
    mid = _METRIC_ID[metric]
    dense_metric = 0 if mid == 1 else mid       # euclidean: ranked on the squared values, square root at the end
    bq = min(_GENERAL_BQ, (nq + 31) // 32 * 32)
    G = torch.empty((bq, min(_GENERAL_BD, (nd + 31) // 32 * 32)), dtype=torch.float32, device=dev)
    for c0 in range(0, nq, _GENERAL_BQ):
        c1 = min(c0 + _GENERAL_BQ, nq)
        for d0 in range(0, nd, _GENERAL_BD):
            d1 = min(d0 + _GENERAL_BD, nd)
            _lib.check(
                L.tdr_dense_dist_packed_f32(_lib.ptr(Q.data[((q0 + c0) // 32) * tile:]), c1 - c0, 0, _lib.ptr(Y.data[(d0 // 32) * tile:]),
                                            d1 - d0, d, dense_metric, 0, _DIAG_ADD, _lib.ptr(G), G.stride(0), st),
                "tdr_dense_dist_packed_f32",
            )
            _lib.check(
                L.tdr_topk_merge_f32(_lib.ptr(G), G.stride(0), c1 - c0, d1 - d0, None, None, q_offset + q0 + c0, d0, k, 3,
                                     1 if exclude_self else 0, _lib.ptr(keys[c0:c1]), st),
                "tdr_topk_merge_f32",
            )
    out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
    _lib.check(L.tdr_topk_emit_f32(_lib.ptr(keys), nq, k, mid, _lib.ptr(out_d), _lib.ptr(out_i), st), "tdr_topk_emit_f32")
    LAST_KNN["path"], LAST_KNN["flagged"] = "dense MFMA blocks + top-k merge", 0
    return out_d, out_i


def _knn_dense_wide(Q: "PackedPoints", Y: "PackedPoints", q0: int, q1: int, k: int, metric: str, exclude_self: bool, q_offset: int):
    """k beyond the running top-k kernel's 1024 entries per query (the reference's ``kmin`` has no limit, utils/utils.py:203-216):
    the SAME exact distance blocks from the dense fp32-MFMA kernel; the running lists are kept by device-side torch sorts in
    the canonical (distance, index) order -- a side path for very wide requests (1024 queries x 65536 rows per block)."""
    L = _lib.lib()
    dev, d = Y.device, Y.d
    nq, nd = q1 - q0, Y.n
    st = _lib.stream_ptr()
    tile = int(L.tdr_packed_floats(32, d))
    mid = _METRIC_ID[metric]
    dense_metric = 0 if mid == 1 else mid
    bq = 1024
    G = torch.empty((bq, min(_GENERAL_BD, (nd + 31) // 32 * 32)), dtype=torch.float32, device=dev)
    out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
    for c0 in range(0, nq, bq):
        c1 = min(c0 + bq, nq)
        run_d = torch.empty((c1 - c0, 0), dtype=torch.float32, device=dev)
        run_i = torch.empty((c1 - c0, 0), dtype=torch.int32, device=dev)
        for d0 in range(0, nd, _GENERAL_BD):
            d1 = min(d0 + _GENERAL_BD, nd)
            _lib.check(
                L.tdr_dense_dist_packed_f32(_lib.ptr(Q.data[((q0 + c0) // 32) * tile:]), c1 - c0, 0, _lib.ptr(Y.data[(d0 // 32) * tile:]),
                                            d1 - d0, d, dense_metric, 0, _DIAG_ADD, _lib.ptr(G), G.stride(0), st),
                "tdr_dense_dist_packed_f32",
            )
            blk = G[: c1 - c0, : d1 - d0]
            if exclude_self:
                own = torch.arange(q_offset + q0 + c0, q_offset + q0 + c1, device=dev) - d0
                hit = (own >= 0) & (own < d1 - d0)
                rows = torch.nonzero(hit).squeeze(1)
                blk = blk.clone()
                blk[rows, own[rows]] = float("inf")
            cand_d = torch.cat([run_d, blk], 1)
            cand_i = torch.cat([run_i, torch.arange(d0, d1, dtype=torch.int32, device=dev).expand(c1 - c0, -1)], 1)
            o = torch.argsort(cand_i, dim=1, stable=True)                 # by index, then (stable) by distance: canonical ties
            cand_d, cand_i = cand_d.gather(1, o), cand_i.gather(1, o)
            o = torch.argsort(cand_d, dim=1, stable=True)[:, :k]
            run_d, run_i = cand_d.gather(1, o), cand_i.gather(1, o)
        out_d[c0:c1], out_i[c0:c1] = (torch.sqrt(run_d) if mid == 1 else run_d), run_i
    LAST_KNN["path"], LAST_KNN["flagged"] = "dense MFMA blocks + torch sort (k > 1024)", 0
    return out_d, out_i


_GENERAL_BQ = 4096    # query rows per library-GEMM block of the general-D path
_GENERAL_BD = 65536   # database rows per block (4096 x 65536 fp32 = 1 GiB)


def _knn_general(Xq, Y, k, metric, exclude_self, q_global0=0):
    """Exact kNN outside the LDS-resident lists of the scan kernels (k beyond their capacity, sqhyperbolic, and the
    manhattan tile pass): per (query chunk x database chunk) block the package's own fp32-MFMA tile kernel forms X Y^T
    (``_gram``; a library GEMM until round 6) and ``tdr_topk_merge_f32`` does the rest of the reference's op sequence (norm
    expansion, self exclusion, running top-k in the canonical (distance, index) order).  D > 256 with an MFMA metric goes
    to ``_knn_wide`` first."""
    L = _lib.lib()
    nq, nd = Xq.shape[0], Y.shape[0]
    if metric in ("sqeuclidean", "euclidean", "angular") and Y.shape[1] > 256 and _opt("WIDE_SCAN"):
        res = _knn_wide(Xq, Y, k, metric, exclude_self, q_global0)
        if res is not None:
            return res
    if k > int(L.tdr_topk_max_k()):
        raise NotImplementedError(f"[torchdr_amd] k={k} > {int(L.tdr_topk_max_k())} is not supported by the running top-k kernel.")
    dev = Y.device
    l1 = metric == "manhattan"
    xn = yn = None
    if not l1:
        xn = (Xq * Xq).sum(1).contiguous()
        yn = xn if Y is Xq else (Y * Y).sum(1).contiguous()
    keys = torch.empty((nq, k), dtype=torch.int64, device=dev)
    st = _lib.stream_ptr()
    _lib.check(L.tdr_topk_init(_lib.ptr(keys), nq, k, st), "tdr_topk_init")
    mid = _METRIC_ID[metric]
    for q0 in range(0, nq, _GENERAL_BQ):
        q1 = min(q0 + _GENERAL_BQ, nq)
        Xc = Xq[q0:q1]
        for d0 in range(0, nd, _GENERAL_BD):
            d1 = min(d0 + _GENERAL_BD, nd)
            G = _l1_block(Xc, Y[d0:d1]) if l1 else _gram(Xc, Y[d0:d1])
            _lib.check(
                L.tdr_topk_merge_f32(_lib.ptr(G), G.stride(0), q1 - q0, d1 - d0, _lib.ptr(None if l1 else xn[q0:q1]),
                                     _lib.ptr(None if l1 else yn[d0:d1]),
                                     q_global0 + q0, d0, k, mid, 1 if exclude_self else 0, _lib.ptr(keys[q0:q1]), st),
                "tdr_topk_merge_f32",
            )
    out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
    _lib.check(L.tdr_topk_emit_f32(_lib.ptr(keys), nq, k, mid, _lib.ptr(out_d), _lib.ptr(out_i), st), "tdr_topk_emit_f32")
    LAST_KNN["path"] = "manhattan (L1 tile kernel + top-k merge)" if l1 else "general-D (MFMA Gram tiles + top-k merge)"
    LAST_KNN["flagged"] = 0
    return out_d, out_i


_L1_MARGIN = 8            # extra candidates kept by the tile pass for the exact re-evaluation
_L1_EXACT_MAX_D = 8192    # tdr_l1_exact_f32 restates two cascade levels of the reference's summation


def _l1_exact(X, q_rows, nq, q_global0, Y, cand, nc, j0, exclude_self):
    """(nq, nc) manhattan distances in the reference's summation order (``tdr_l1_exact_f32``)."""
    out = torch.empty((nq, nc), dtype=torch.float32, device=Y.device)
    _lib.check(
        _lib.lib().tdr_l1_exact_f32(_lib.ptr(X), X.stride(0), _lib.ptr(q_rows), nq, q_global0, _lib.ptr(Y), Y.stride(0),
                                    _lib.ptr(cand), 0 if cand is None else cand.stride(0), nc, j0, X.shape[1],
                                    1 if exclude_self else 0, _lib.ptr(out), out.stride(0), _lib.stream_ptr()),
        "tdr_l1_exact_f32",
    )
    return out


def _rank_candidates(E, cand, k):
    """Rows of (distance, index) candidates -> the k smallest by (distance, index)."""
    L = _lib.lib()
    nq = E.shape[0]
    st = _lib.stream_ptr()
    keys = torch.empty((nq, k), dtype=torch.int64, device=E.device)
    _lib.check(L.tdr_topk_init(_lib.ptr(keys), nq, k, st), "tdr_topk_init")
    _lib.check(L.tdr_topk_merge_cand_f32(_lib.ptr(E), _lib.ptr(cand), E.stride(0), nq, E.shape[1], k, _lib.ptr(keys), st),
               "tdr_topk_merge_cand_f32")
    out_d = torch.empty((nq, k), dtype=torch.float32, device=E.device)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=E.device)
    _lib.check(L.tdr_topk_emit_f32(_lib.ptr(keys), nq, k, 3, _lib.ptr(out_d), _lib.ptr(out_i), st), "tdr_topk_emit_f32")
    return out_d, out_i


def _knn_manhattan(Xq, Y, k, exclude_self, q_global0=0):
    """Exact manhattan kNN whose values and indices are the reference CPU backend's (distance/torch.py:96-98, 118-120).

    Two passes, like the Euclidean search: (1) the VALU tile kernel (fast, its own summation order) keeps the k + 8 best
    per query; (2) those candidates are re-evaluated in the reference's summation order and ranked by (distance,
    index).  Both are fp32 sums of the SAME non-negative terms fl(x - y), so they differ by at most
    ``delta = 2.5 d 2^-24`` relatively; a point outside the candidate list has tile distance >= a_L, hence reference
    distance >= a_L (1 - delta), while the reference's k-th distance is <= a_k (1 + delta): when
    a_L (1 - delta) > a_k (1 + delta) nothing outside the list can enter the top k.  Rows failing that certificate
    (heavy ties / duplicates) are re-searched against the whole database in the reference's order."""
    nq, nd, d = Xq.shape[0], Y.shape[0], Xq.shape[1]
    n_valid = nd - (1 if exclude_self else 0)
    if d >= _L1_EXACT_MAX_D:  # a third cascade level in the reference's sum: tile-order values (fp32 rounding apart)
        return _knn_general(Xq, Y, k, "manhattan", exclude_self, q_global0)
    L = min(k + _L1_MARGIN, n_valid, int(_lib.lib().tdr_topk_max_k()))
    Ca, Ia = _knn_general(Xq, Y, L, "manhattan", exclude_self, q_global0)
    E = _l1_exact(Xq, None, nq, q_global0, Y, Ia, L, 0, False)
    C, I = _rank_candidates(E, Ia, k)
    n_bad = 0
    if L < n_valid:
        delta = 2.5 * d * 2.0 ** -24
        bad = torch.nonzero(Ca[:, L - 1] * (1.0 - delta) <= Ca[:, k - 1] * (1.0 + delta)).reshape(-1)
        n_bad = int(bad.numel())
        for b0 in range(0, n_bad, 2048):
            rows = bad[b0:b0 + 2048].contiguous()
            m = int(rows.numel())
            keys = torch.empty((m, k), dtype=torch.int64, device=Y.device)
            st = _lib.stream_ptr()
            _lib.check(_lib.lib().tdr_topk_init(_lib.ptr(keys), m, k, st), "tdr_topk_init")
            for d0 in range(0, nd, _GENERAL_BD):
                d1 = min(d0 + _GENERAL_BD, nd)
                Ef = _l1_exact(Xq, rows, m, q_global0, Y, None, d1 - d0, d0, exclude_self)  # own row -> +inf
                _lib.check(
                    _lib.lib().tdr_topk_merge_f32(_lib.ptr(Ef), Ef.stride(0), m, d1 - d0, _lib.ptr(None), _lib.ptr(None), 0,
                                                  d0, k, 3, 0, _lib.ptr(keys), st),
                    "tdr_topk_merge_f32",
                )
            od = torch.empty((m, k), dtype=torch.float32, device=Y.device)
            oi = torch.empty((m, k), dtype=torch.int32, device=Y.device)
            _lib.check(_lib.lib().tdr_topk_emit_f32(_lib.ptr(keys), m, k, 3, _lib.ptr(od), _lib.ptr(oi), st), "tdr_topk_emit_f32")
            C[rows] = od
            I[rows] = oi
    LAST_KNN["path"], LAST_KNN["flagged"] = "manhattan (L1 tile kernel + exact-order re-evaluation)", n_bad
    return C, I


def _l1_block(X, Y):
    """(nq, nd) block of manhattan distances (distance/torch.py:96-98) by the VALU tile kernel."""
    out = torch.empty((X.shape[0], Y.shape[0]), dtype=torch.float32, device=X.device)
    _lib.check(
        _lib.lib().tdr_l1_block_f32(_lib.ptr(X), X.stride(0), X.shape[0], _lib.ptr(Y), Y.stride(0), Y.shape[0], X.shape[1],
                                    _lib.ptr(out), out.stride(0), _lib.stream_ptr()),
        "tdr_l1_block_f32",
    )
    return out


def _gram(X, Y):
    """X Y^T by the package's own fp32-MFMA tile kernels (the "angular" epilogue is -x.y): one k-ordered fma chain per entry -- what
    the reference's sgemm computes for D <= ~380 -- instead of a library GEMM (round 6: the last two `torch.mm` call sites on the
    path -- the sqhyperbolic forms and the comparison form of the general-D search -- went through rocBLAS)."""
    if X.shape[1] <= 256:
        return dense_packed(PackedPoints(X), PackedPoints(Y), "angular", False).neg_()
    Yp = WidePackedPoints(Y)
    Qp = Yp if X is Y else WidePackedPoints(X)
    out = torch.empty((Qp.n, Yp.n), dtype=torch.float32, device=Yp.device)
    _lib.check(
        _lib.lib().tdr_dense_dist_wide_f32(_lib.ptr(Qp.data), Qp.n, 0, _lib.ptr(Yp.data), Yp.n, Yp.d, _METRIC_ID["angular"], 0, _DIAG_ADD,
                                           _lib.ptr(out), out.stride(0), _lib.stream_ptr()),
        "tdr_dense_dist_wide_f32",
    )
    return out.neg_()


def _dense_general(X, Y, metric, exclude_self):
    """Dense distance matrix outside the register-resident kernels (distance/torch.py:82-116): D > 256 on the wide tile
    kernel, sqhyperbolic with a library GEMM, manhattan on the L1 kernel."""
    if metric == "manhattan":
        C = _l1_block(X, Y)
        if exclude_self:
            C.diagonal().add_(_DIAG_ADD)
        return C
    if metric in ("sqeuclidean", "euclidean", "angular") and X.shape[1] > 256 and _opt("WIDE_SCAN") and Y.shape[0] <= 65535 * 32:
        # the wide tile images and one fp32-MFMA tile kernel (no library GEMM)
        Yp = WidePackedPoints(Y)
        Qp = Yp if X is Y else WidePackedPoints(X)
        out = torch.empty((Qp.n, Yp.n), dtype=torch.float32, device=Yp.device)
        _lib.check(
            _lib.lib().tdr_dense_dist_wide_f32(_lib.ptr(Qp.data), Qp.n, 0, _lib.ptr(Yp.data), Yp.n, Yp.d, _METRIC_ID[metric],
                                               1 if exclude_self else 0, _DIAG_ADD, _lib.ptr(out), out.stride(0), _lib.stream_ptr()),
            "tdr_dense_dist_wide_f32",
        )
        return out
    G = _gram(X, Y) if Y.shape[0] <= 65535 * 32 else torch.mm(X, Y.t())
    if metric == "sqhyperbolic":  # distance/torch.py:101-107
        xn, yn = (X * X).sum(1).contiguous(), (Y * Y).sum(1).contiguous()
        for r0 in range(0, G.shape[0], 32768):
            r1 = min(r0 + 32768, G.shape[0])
            _lib.check(
                _lib.lib().tdr_hyperbolic_from_gram_f32(_lib.ptr(G[r0:r1]), G.stride(0), r1 - r0, G.shape[1],
                                                        _lib.ptr(xn[r0:r1]), _lib.ptr(yn), _lib.stream_ptr()),
                "tdr_hyperbolic_from_gram_f32",
            )
        if exclude_self:
            G.diagonal().add_(_DIAG_ADD)
        return G
    if metric == "angular":
        C = -G
    else:
        C = ((X * X).sum(1)[:, None] + (Y * Y).sum(1)[None, :]) - 2.0 * G
        if metric == "euclidean":
            C = C.clamp_(min=0).sqrt_()
    if exclude_self:
        C.diagonal().add_(_DIAG_ADD)
    return C


def pairwise_distances(
    X: torch.Tensor,
    Y: Optional[torch.Tensor] = None,
    metric: str = "euclidean",
    backend=None,
    exclude_diag: bool = False,
    k: Optional[int] = None,
    return_indices: bool = False,
    device: str = "auto",
    distributed_ctx: Optional[DistributedContext] = None,
):
    r"""Compute pairwise distances (or the k smallest per row) between two point sets.

    (Body: :func:`_pairwise`, which additionally takes the ``info`` record the affinity classes use.)

    Same contract as the reference (``distance/base.py:22-249``, ``distance/torch.py:21-125``):
    ``C_ij = (||x_i||^2 + ||y_j||^2) - 2 x_i.y_j`` in fp32 (``sqeuclidean``, not clamped),
    ``sqrt(clamp(C, 0))`` (``euclidean``), ``-x_i.y_j`` (``angular``); when ``Y`` is None / is X
    and ``exclude_diag`` the self distance is excluded (``+1e12`` on the diagonal of the dense
    matrix); with ``k`` the k smallest per row, ascending, indices int32; ``k >= n_columns``
    returns the dense matrix and ``None`` indices (``utils/utils.py:203-204``).  With a
    ``distributed_ctx`` each rank searches its row chunk against the full database
    (``distance/base.py:160-211``).

    Rows are ordered by (distance, index); ``torch.topk``'s order among exactly tied
    distances is unspecified, so that is the only place results can differ from the
    reference's CPU backend (parity protocol).
    """
    return _pairwise(X, Y, metric, backend, exclude_diag, k, return_indices, device, distributed_ctx, None)


def _pairwise(X, Y, metric, backend, exclude_diag, k, return_indices, device, distributed_ctx, info):
    """:func:`pairwise_distances` with an ``info`` record (dict or None) between the search and the affinity class that
    called it -- in: ``want_loop_order`` (a row-sharded UMAP accepts its rows in the cluster-sorted numbering); out:
    ``cluster_order`` = (perm, inv) of a pruned / approximate self search, ``loop_order`` = True when the returned rows
    ARE in that numbering.  (Replaces the module-level record round 2 passed this through.)"""
    faiss_like = backend == "faiss" or type(backend).__name__ == "FaissConfig"
    if metric not in LIST_METRICS:
        if faiss_like:   # the message of the reference's Faiss backend (distance/faiss.py:297-300)
            raise ValueError("[TorchDR] Only ['euclidean', 'sqeuclidean', 'angular'] metrics are supported for FAISS.")
        raise ValueError(f"[TorchDR] ERROR : The '{metric}' distance is not supported.")
    if hasattr(backend, "check_index_type"):
        backend.check_index_type()
    if k is not None and not isinstance(k, int):   # a 0-d tensor or a numpy integer (reference tests/test_utils.py:165-174)
        k = int(k)
    if getattr(backend, "index_type", None) == "IVFPQ" and isinstance(X, torch.Tensor) and X.dim() == 2:
        # there are no sub-quantisers here (an IVFPQ request is served uncompressed), but a configuration Faiss would
        # refuse is refused in the same words (distance/faiss.py:341-347)
        d_feat, M = X.shape[1], int(getattr(backend, "M", 16))
        if M <= 0 or d_feat % M != 0:
            raise ValueError(
                f"[TorchDR] ERROR : Vector dimension {d_feat} must be divisible by M={M} for IVFPQ. "
                f"Choose M from divisors of {d_feat}."
            )
    if is_dataloader(X):  # reference base.py:121-157 (batches -> one HBM-resident tensor, utils/dataloader.py)
        if k is None:
            raise ValueError(
                "[TorchDR] DataLoader input requires k-NN computation. k cannot be None when X is a DataLoader."
            )
        if Y is not None:
            raise ValueError(
                "[TorchDR] DataLoader input does not support cross-distance. Y must be None when X is a DataLoader."
            )
        if not (backend is None or backend == "faiss" or type(backend).__name__ == "FaissConfig"):
            raise ValueError(
                f"[TorchDR] DataLoader input only supports FAISS backend, got backend='{backend}'. "
                "Use backend='faiss' or backend=None."
            )
        from torchdr_amd.utils.dataloader import stream_dataloader_packed

        X, packed_stream = stream_dataloader_packed(X, None if device == "auto" else device, metric)
    else:
        packed_stream = None
    if not isinstance(X, torch.Tensor):
        raise TypeError("[torchdr_amd] pairwise_distances expects a torch.Tensor or a DataLoader.")

    self_search = Y is None or Y is X
    if X.dtype == torch.float64:
        res = _pairwise_f64(X, None if self_search else Y, metric, exclude_diag, k, return_indices, device, distributed_ctx)
        if res is not NotImplemented:
            return res
        # shapes / metrics without a float64 kernel (d > 256, manhattan, sqhyperbolic, row-sharded): float32 arithmetic
        out = pairwise_distances(as_float32(X), None if self_search else as_float32(Y), metric=metric, backend=backend,
                                 exclude_diag=exclude_diag, k=k, return_indices=return_indices, device=device,
                                 distributed_ctx=distributed_ctx)
        if isinstance(out, tuple):
            return out[0].to(torch.float64), out[1]
        return out.to(torch.float64)
    X = _to_device(X, device)
    _lib.require_gpu(X, "X")
    if not self_search:
        Y = _to_device(Y, device)
        _lib.require_gpu(Y, "Y")

    # --- distributed: queries = this rank's chunk, database = full X (base.py:160-211)
    if distributed_ctx is not None and distributed_ctx.is_initialized:
        if k is None:
            raise ValueError(
                "[TorchDR] Distributed mode requires sparse computation with k-NN. "
                "k cannot be None when distributed_ctx is provided."
            )
        if Y is not None:
            raise ValueError(
                "[TorchDR] Distributed mode does not support cross-distance computation. "
                "Y must be None when distributed_ctx is provided."
            )
        n = X.shape[0]
        c0, c1 = distributed_ctx.compute_chunk_bounds(n)
        if X.shape[1] > 256 or metric in _GENERAL_ONLY:
            if X.dtype != torch.float32:
                raise NotImplementedError(f"[torchdr_amd] only float32 inputs are supported by the HIP distance kernels (got {X.dtype}).")
            Xc = X if X.stride(1) == 1 else X.contiguous()
            if metric == "manhattan":
                C, I = _knn_manhattan(Xc[c0:c1], Xc, int(k), bool(exclude_diag), q_global0=c0)
            else:
                C, I = _knn_general(Xc[c0:c1], Xc, int(k), metric, bool(exclude_diag), q_global0=c0)
            return (C, I) if return_indices else C
        Yp = PackedPoints(X)
        # every rank must take the same branch (the sharded search is collective): decide from rank-independent
        # quantities only (n // W, never this rank's own c1 - c0, which differs by one row across ranks)
        if (distributed_ctx.world_size > 1 and _want_prune(Yp, n)
                and _use_screen(Yp, Yp, n // distributed_ctx.world_size, int(k), metric)
                and n // distributed_ctx.world_size >= _SCREEN_PILOT_Q):
            res = knn_pruned_sharded(Yp, int(k), metric, bool(exclude_diag), distributed_ctx,
                                     loop_order=bool(info and info.get("want_loop_order")), info=info)
            if res is not None:
                return res if return_indices else res[0]
        Qp = Yp if c0 % 32 == 0 else PackedPoints(X[c0:c1])
        rows = slice(c0, c1) if Qp is Yp else None
        # the reference asks Faiss for k+1 and drops column 0; excluding the query's own row
        # by index is the same thing whenever the point is its own strict nearest neighbour.
        C, I = knn_packed(
            Qp, Yp, k, metric, exclude_self=exclude_diag,
            q_offset=0 if Qp is Yp else c0, q_rows=rows,
        )
        return (C, I) if return_indices else C

    do_exclude = bool(exclude_diag) and self_search
    if X.shape[1] > 256 or metric in _GENERAL_ONLY:  # not on the MFMA scan kernels: (library GEMM | L1 tiles) + HIP top-k merge
        if X.dtype != torch.float32:
            raise NotImplementedError(f"[torchdr_amd] only float32 inputs are supported by the HIP distance kernels (got {X.dtype}).")
        Xc = X if X.stride(1) == 1 else X.contiguous()
        Yc = Xc if self_search else (Y if Y.stride(1) == 1 else Y.contiguous())
        if Xc.shape[1] != Yc.shape[1]:
            raise ValueError("[TorchDR] ERROR : X and Y must have the same number of features.")
        if k is not None and k < Yc.shape[0]:
            if do_exclude and k > Yc.shape[0] - 1:
                raise ValueError("[TorchDR] ERROR : k must be smaller than the number of samples.")
            if metric == "manhattan":
                C, I = _knn_manhattan(Xc, Yc, int(k), do_exclude)
            else:
                C, I = _knn_general(Xc, Yc, int(k), metric, do_exclude)
            return (C, I) if return_indices else C
        C = _dense_general(Xc, Yc, metric, do_exclude)
        return (C, None) if return_indices else C
    Xp = packed_stream if packed_stream is not None and packed_stream.X is X else PackedPoints(X)
    Yp = Xp if self_search else PackedPoints(Y)
    n_cols = Yp.n

    if k is not None and k < n_cols:
        if do_exclude and k > n_cols - 1:
            raise ValueError("[TorchDR] ERROR : k must be smaller than the number of samples.")
        ivf = _ivf_request(backend, self_search, k, n_cols, Xp.d, metric)
        if ivf is not None:   # FaissConfig(index_type="IVF" / "IVFPQ"): approximate search on the cluster index
            C, I = _knn_ivf(Yp, int(k), metric, do_exclude, *ivf, info=info)
        else:
            C, I = knn_packed(Xp, Yp, int(k), metric, do_exclude, info=info)
        return (C, I) if return_indices else C

    C = dense_packed(Xp, Yp, metric, do_exclude)
    if return_indices:
        return C, None
    return C


def pairwise_distances_indexed(
    X: torch.Tensor,
    query_indices: Optional[torch.Tensor] = None,
    key_indices: Optional[torch.Tensor] = None,
    Y: Optional[torch.Tensor] = None,
    metric: str = "sqeuclidean",
    backend=None,
    device: str = "auto",
):
    r"""Distances between indexed subsets (reference ``distance/base.py:252-405``).

    The per-query-keys form (``key_indices`` 2-D, the one the embedding loop uses) runs the
    gather kernel ``tdr_indexed_sqdist_f32``: ``sum_c (x_ic - y_jc)^2`` by direct difference
    (``base.py:384-385``; ``sum |.|`` for manhattan, ``-sum x y`` for angular, ``:388-391``).  A key index of -1 wraps to the last row, as PyTorch indexing does
    in the reference.
    """
    if Y is None:
        Y = X
    X = _to_device(X, device)
    Y = _to_device(Y, device)
    _lib.require_gpu(X, "X")
    if metric not in LIST_METRICS:
        raise NotImplementedError(f"Metric '{metric}' not implemented for indexed distances")
    if query_indices is not None and query_indices.dim() != 1:
        raise NotImplementedError("2D query indices not yet supported")
    if key_indices is not None and key_indices.dim() not in (1, 2):
        raise ValueError(f"key_indices must be 1D or 2D, got {key_indices.dim()}D")
    if key_indices is None or key_indices.dim() == 1:
        # queries x keys block (reference base.py:357-376, torch.cdist): gather the rows, dense MFMA kernel
        Xq = X if query_indices is None else X[query_indices.to(X.device).long()]
        Yk = Y if key_indices is None else Y[key_indices.to(Y.device).long()]
        Xq, Yk = Xq.float().contiguous(), Yk.float().contiguous()
        if metric in _GENERAL_ONLY or Xq.shape[1] > 256:
            return _dense_general(Xq, Yk, metric, False)
        return dense_packed(PackedPoints(Xq), PackedPoints(Yk), metric, False)
    L = _lib.lib()
    f64 = X.dtype == torch.float64 and metric != "sqhyperbolic"
    Xc = X.contiguous() if f64 else X.contiguous().float()
    Yc = Y.contiguous().to(torch.float64) if f64 else Y.contiguous().float()
    keys = key_indices.to(device=X.device, dtype=torch.int64).contiguous()
    nq = keys.shape[0]
    if query_indices is None:
        q = torch.arange(nq, device=X.device, dtype=torch.int64)
    else:
        q = query_indices.to(device=X.device, dtype=torch.int64).contiguous()
        assert keys.shape[0] == len(q), (
            f"key_indices first dim {keys.shape[0]} must match number of queries {len(q)}"
        )
    if f64:   # float64 inputs: float64 arithmetic, as the reference's gather-and-subtract does (distance/base.py:384-391)
        out = torch.empty(keys.shape, dtype=torch.float64, device=X.device)
        _lib.check(L.tdr_indexed_sqdist_f64(_lib.ptr(Xc), Xc.shape[0], Xc.shape[1], _lib.ptr(Yc), Yc.shape[0], _lib.ptr(q), nq,
                                            keys.shape[1], {"sqeuclidean": 0, "euclidean": 1, "manhattan": 2, "angular": 3}[metric],
                                            _lib.ptr(keys), _lib.ptr(out), _lib.stream_ptr()), "tdr_indexed_sqdist_f64")
        return out
    out = torch.empty(keys.shape, dtype=torch.float32, device=X.device)
    _lib.check(
        L.tdr_indexed_sqdist_f32(
            _lib.ptr(Xc), Xc.shape[0], Xc.shape[1], _lib.ptr(Yc), Yc.shape[0], _lib.ptr(q), nq,
            keys.shape[1], {"sqeuclidean": 0, "euclidean": 1, "manhattan": 2, "angular": 3, "sqhyperbolic": 4}[metric], _lib.ptr(keys),
            _lib.ptr(out),
            _lib.stream_ptr(),
        ),
        "tdr_indexed_sqdist_f32",
    )
    return out
