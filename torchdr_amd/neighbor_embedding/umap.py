"""UMAP on MI355X -- mirror of ``torchdr/neighbor_embedding/umap.py`` (reference :19-292)."""

import functools
from typing import Dict, Optional, Type, Union

import numpy as np
import torch

from torchdr_amd import _lib
from torchdr_amd.affinity import UMAPAffinity
from torchdr_amd.neighbor_embedding.base import NegativeSamplingNeighborEmbedding
from torchdr_amd.utils.sparse import CSRAffinity


# bench.py sets this to a list to collect ("grad", start_event, end_event, nnz) around every PROFILE_EVERY-th gradient
# evaluation (iterations PROFILE_EVERY // 2, + PROFILE_EVERY, ...) and ("build", start_event, end_event, n_iters) around every schedule build (HIP events on the launch
# stream); None = no instrumentation.
PROFILE = None
PROFILE_EVERY = 25
PROFILE_KEEP_GRAD = False     # True: the fused combine + step kernel writes the gradient at every iteration (debugging)

# scheduled loop (csrc/tdr_umap_sched.hip): the epoch counters are advanced SCHED_BLOCK_ITERS iterations at a time and
# the gradient kernel reads per-iteration lists of the edges that fire.  SCHEDULED = False selects the per-step kernel
# (tdr_umap_grad_f32), which streams every edge's counter each iteration; SCHED_GEOM: low 4 bits = lane geometry of the
# gradient kernel (tools/umap_sched_perf.py), bit 4 = all L2 slices in ONE launch spread over the XCDs (0.295 vs 0.304 ms per
# iteration at N = 1M, bit-identical gradients); SCHED_SLICES overrides the automatic number of L2 slices of the embedding.
SCHEDULED = True
# LOOP_RUNNER: run the whole optimisation loop through tdr_umap_loop_* (one ctypes call per window of <= 32 iterations,
# windows replayed as HIP graphs when LOOP_GRAPH) when the estimator's step is the stock one (no overridden hooks, no
# injected negatives, fused SGD, no early exaggeration).  "auto": only where the host would otherwise bound the loop,
# i.e. row-sharded runs with an on-stream RCCL context (an iteration there is a few tens of microseconds of kernels);
# on one GPU an iteration is ~0.37 ms of kernels at N = 1M and the ~50 us of Python per iteration are hidden -- measured
# (bench.py --loop): python 475.7 ms / step, C loop 486.4, graphs 496.3 (capture + 129-node replays cost more than the
# launches they save).  True: whenever eligible; False: never.
LOOP_RUNNER = "auto"
LOOP_GRAPH = True
# bench.py sets this to a list to collect (start_event, end_event, n_iterations) per loop-runner segment
LOOP_PROFILE = None
MERGED_CHECK = True      # single process: the NaN flag, the schedule's error words and |grad| of a check iteration come back in ONE host read
SCHED_BLOCK_ITERS = 32
# (round 5: the variants that were measured slower and kept behind switches -- rows of a workgroup dealt by load (geom bit 6),
# the SGD step fused into the gradient launch, the schedule built one window ahead on a side stream -- are gone from the library;
# their measurements stay in profiles/r04_grad_ablation.json, r04_fused_step.json, r04_build_ahead.json)
SCHED_GEOM = 16
SCHED_SLICES = 0
# RELABEL: when the kNN stage worked in a cluster-sorted row order (pruned search) and nothing outside this class looks at
# rows during the optimisation (stock hooks, one GPU, no neighbour exclusion), the loop numbers the points in that order --
# a row's neighbours then sit in the same few cache lines of the embedding, and the gathers of the fired edges (16 % of
# all gathers; the negatives are uniform by definition) mostly hit the L1.  0.300 -> 0.274 ms per iteration at N = 1M
# together with the row sort above.  The embedding is returned in the caller's order; the negative sampler is keyed by
# the loop's row numbers, so the random stream differs from an unrelabelled run (same distribution).
RELABEL = True
# GROUPED: the scheduled loop keeps its per-edge state (columns, epochs_per_sample, epoch_of_next_sample) in GROUP order --
# the edges of every 16 consecutive rows sorted by firing-period class (tdr_umap_sched_group_f32) -- and builds the firing
# lists with tdr_umap_sched_build_groups_f32 (one wavefront per group, equally busy lanes, coalesced list stores).  The
# row-major arrays stay what they were (`epochs_per_sample`, `_loop_cols`); `epoch_of_next_sample` is brought back to the
# row-major order when it is read.  False: the row-chunk kernel of rounds 2-3 (tdr_umap_sched_build_f32).
GROUPED = True
SCHED_STAGE = 0      # LDS stage entries of the grouped build (0 = library default)
# NEGATIVES: where the scheduled loop's negatives come from (neighbor_embedding/base.py:617-649).
#   "pool": per iteration every block of POOL rows stages runs of 16 consecutive rows of Z -- each run uniform over the data
#           set -- into LDS and its rows draw their negatives from that pool (csrc/tdr_umap_pool.hip): uniform marginal law,
#           rows of a block share the iteration's pool; one L2 request per staged LINE instead of one per negative, no L2
#           slices, one lane per row.  float32, 2 or 3 components, no injected / excluded negatives; anything else falls
#           back to "iid" on the same (one-slice) lists.
#   "iid" : every negative an independent uniform draw gathered from L2 (csrc/tdr_umap_sched.hip): the reference's joint
#           law -- the parity path (injected-negative tests, `discard_NNs`).
NEGATIVES = "pool"
POOL_GEOM = 0        # geometry of the pool kernel (0 = library default; tuning knob, changes the sampler's stream)
POOL_FUSED_STEP = True   # torch.optim.SGD's step inside the pool gradient launch (stock step, no momentum), two embedding buffers

def _opt(name):
    """A behaviour switch of this module: the scoped override (torchdr_amd.config.options) or the module attribute."""
    from torchdr_amd import config

    return config.get(name, globals())



def _stock_start():
    from torchdr_amd.neighbor_embedding.base import NegativeSamplingNeighborEmbedding

    return NegativeSamplingNeighborEmbedding.on_training_step_start


@functools.lru_cache(maxsize=64)
def find_ab_params(spread, min_dist):
    """Fit a, b of 1/(1 + a x^(2b)) to the smooth-step target curve (reference :19-36: same grid,
    scipy ``curve_fit``; defaults give a = 1.5769..., b = 0.8950...).  A pure function of its two arguments: the fit
    (0.5 ms of scipy per constructor call) is kept per (spread, min_dist)."""
    from scipy.optimize import curve_fit

    def curve(x, a, b):
        return 1.0 / (1.0 + a * x ** (2 * b))

    xv = np.linspace(0, spread * 3, 300)
    yv = np.zeros(xv.shape)
    yv[xv < min_dist] = 1.0
    yv[xv >= min_dist] = np.exp(-(xv[xv >= min_dist] - min_dist) / spread)
    params, _ = curve_fit(curve, xv, yv)
    return params[0].item(), params[1].item()


class UMAP(NegativeSamplingNeighborEmbedding):
    """UMAP with the reference's constructor (``umap.py:129-160``) and semantics (appendix A.3 of
    SURVEY.md): squared-Euclidean kNN, sigma search, fuzzy-union symmetrisation, per-edge epoch
    counters, 5 negatives per active edge, forces clamped to [-4, 4], plain SGD with a linear
    1 -> 0 learning-rate ramp.  float64 inputs run in float64 end to end (per-step kernel
    ``tdr_umap_grad_f64``; the scheduled loop is the float32 path)."""

    _float64_loop = True

    def __init__(self, n_neighbors: float = 30, n_components: int = 2, min_dist: float = 0.1, spread: float = 1.0,
                 a: Optional[float] = None, b: Optional[float] = None, lr: float = 1e0,
                 optimizer: Union[str, Type[torch.optim.Optimizer]] = "SGD",
                 optimizer_kwargs: Union[Dict, str] = None,
                 scheduler: Optional[Union[str, Type[torch.optim.lr_scheduler.LRScheduler]]] = "LinearLR",
                 scheduler_kwargs: Union[Dict, str, None] = "auto", init: str = "pca", init_scaling: float = 1e-4,
                 min_grad_norm: float = 1e-7, max_iter: int = 1000, device: str = "auto", backend="faiss",
                 verbose: bool = False, random_state: Optional[float] = None, max_iter_affinity: int = 100,
                 metric: str = "sqeuclidean", negative_sample_rate: int = 5, check_interval: int = 50,
                 discard_NNs: bool = False, compile: bool = False, distributed: Union[bool, str] = "auto",
                 **kwargs):
        self.n_neighbors = n_neighbors
        self.min_dist = min_dist
        self.spread = spread
        self.metric = metric
        self.max_iter_affinity = max_iter_affinity
        self.negative_sample_rate = negative_sample_rate
        self.sparsity = True
        self._use_closed_form_gradients = True
        self._eps = 1e-3
        self.a, self.b = a, b  # as given (sklearn get_params); the fitted curve parameters live in _a / _b
        if a is None or b is None:
            a, b = find_ab_params(self.spread, self.min_dist)
        self._a = a
        self._b = b
        self.n_negatives = int(self.negative_sample_rate * self.n_neighbors)
        affinity_in = UMAPAffinity(n_neighbors=n_neighbors, metric=metric, max_iter=max_iter_affinity,
                                   device=device, backend=backend, verbose=verbose, sparsity=self.sparsity,
                                   compile=compile, distributed=distributed)
        super().__init__(affinity_in=affinity_in, n_components=n_components, optimizer=optimizer,
                         optimizer_kwargs=optimizer_kwargs, min_grad_norm=min_grad_norm, max_iter=max_iter, lr=lr,
                         scheduler=scheduler, scheduler_kwargs=scheduler_kwargs, init=init,
                         init_scaling=init_scaling, device=device, backend=backend, verbose=verbose,
                         random_state=random_state, check_interval=check_interval, discard_NNs=discard_NNs,
                         compile=compile, n_negatives=self.n_negatives, distributed=distributed, **kwargs)

    # epoch_of_next_sample (umap.py:232,247) in the row-major loop order.  While the grouped build owns the counters
    # (self._g), they live in group order and are brought back here when somebody looks.
    @property
    def epoch_of_next_sample(self):
        if "_next_rm" not in self.__dict__:
            raise AttributeError("epoch_of_next_sample")
        g = self.__dict__.get("_g")
        if g is not None and g["dirty"]:
            _lib.check(_lib.lib().tdr_umap_sched_ungroup_f32(_lib.ptr(self._csr_loop.rowptr), _lib.ptr(g["order"]), _lib.ptr(g["next"]),
                                                              self._csr_loop.n, _lib.ptr(self._next_rm), _lib.stream_ptr()),
                       "tdr_umap_sched_ungroup_f32")
            g["dirty"] = False
        return self._next_rm

    @epoch_of_next_sample.setter
    def epoch_of_next_sample(self, value):
        self._next_rm = value
        if self.__dict__.get("_g") is not None:   # a caller replaced the counters: the grouped copy follows
            g = self._g
            g0 = self._csr_loop.rowptr[:-1:16]
            e0 = torch.repeat_interleave(g0, torch.cat([g0[1:], self._csr_loop.rowptr[-1:]]) - g0)
            g["next"] = value[e0 + g["order"].long()].contiguous()
            g["dirty"] = False

    @epoch_of_next_sample.deleter
    def epoch_of_next_sample(self):
        self.__dict__.pop("_next_rm", None)

    def _sched_slices(self) -> int:
        if self._pool_negatives():
            return 1    # negatives come from LDS and the fired edges are local in the loop's numbering: nothing to slice
        return int(_opt("SCHED_SLICES")) or int(_lib.lib().tdr_umap_sched_slices(self.n_samples_in_, self.n_components))

    def _pool_negatives(self) -> bool:
        """Decided once per fit (the loop layout depends on it): the pool sampler serves this fit."""
        p = self.__dict__.get("_pool")
        if p is None:
            L = _lib.lib()
            p = bool(_opt("NEGATIVES") == "pool" and _opt("SCHEDULED") and not _opt("SCHED_SLICES") and not self.discard_NNs
                     and self.neg_indices_ is None and L.tdr_umap_pool_supported(int(self.n_components))
                     and self.n_samples_in_ * int(self.n_components) * 4 < 2**32 - 1)
            self._pool = p
        return p

    # the affinity stays in CSR on the device (the reference's padded (N, max_deg) layout is 5-8x larger)
    def _compute_affinity_in(self, X):
        # row-sharded: a fit whose loop may run in the cluster-sorted numbering tells the affinity so BEFORE the search --
        # each rank then keeps the rows of its range of that order (no row exchange, neighbours mostly rank-local).  The
        # test uses nothing rank-dependent: every rank takes the same branch.
        self.affinity_in._accept_loop_order = bool(self.world_size > 1 and self._relabel_eligible_static())
        self._csr = self.affinity_in(X, return_indices=True, return_csr=True)

    def _relabel_eligible(self) -> bool:
        if not self._relabel_eligible_static():
            return False
        if getattr(self.affinity_in, "_rows_in_loop_order", False):
            return True     # row-sharded: the affinity already IS in the order
        return self.world_size == 1 and self._csr.vals.dtype == torch.float32 and self._csr.n == self._csr.n_total

    def _relabel_eligible_static(self) -> bool:
        """What can be said before the affinity exists: the switches, and that nothing outside this class looks at rows
        during the optimisation."""
        from torchdr_amd.affinity_matcher import AffinityMatcher
        from torchdr_amd.neighbor_embedding.base import NegativeSamplingNeighborEmbedding, NeighborEmbedding

        if not (_opt("RELABEL") and _opt("SCHEDULED")) or self.discard_NNs or self.neg_indices_ is not None:
            return False
        if self.n_samples_in_ >= 2**31 - 1:
            return False
        cls = type(self)
        stock = (
            ("on_training_step_start", NegativeSamplingNeighborEmbedding), ("on_training_step_end", NeighborEmbedding),
            ("_training_step", AffinityMatcher), ("_optimizer_step", AffinityMatcher), ("_sgd_kernel", UMAP),
            ("_compute_gradients", UMAP), ("_compute_gradients_scheduled", UMAP), ("_grad_norm", UMAP),
            ("_init_embedding", NeighborEmbedding), ("_run_training_loop", UMAP), ("_loop_segments", UMAP), ("_fit_transform", UMAP),
            ("on_affinity_computation_end", UMAP), ("_compute_affinity_in", UMAP), ("_converged", AffinityMatcher),
        )
        return all(getattr(cls, name) is getattr(owner, name) for name, owner in stock)

    def _relabel(self):
        """The loop's copy of the graph, numbered in the kNN stage's cluster-sorted order (``tdr_csr_permute_f32``);
        ``self._perm[j]`` = caller's row of loop row j.  Returns the graph the loop runs on.  The order comes from the
        cluster index (members of a cluster by ascending row: the same on every run and on every rank -- the loop's
        numbering, and with it the negative sampler, must not depend on arrival order).  Row-sharded fits receive their
        affinity rows in that numbering already (``UMAPAffinity._rows_in_loop_order``): only the map is kept."""
        order = getattr(self.affinity_in, "_row_order", None)
        self.affinity_in._row_order = None
        if order is None or not self._relabel_eligible():
            if getattr(self.affinity_in, "_rows_in_loop_order", False):
                raise RuntimeError("[torchdr_amd] UMAP: the affinity rows are in cluster order but the loop cannot run in it.")
            return self._csr
        perm, inv = order
        csr, n, dev = self._csr, self._csr.n_total, self._csr.vals.device
        if perm.numel() != n:
            return csr
        self._perm = self.loop_order_ = perm.to(torch.int64)
        if getattr(self.affinity_in, "_rows_in_loop_order", False):
            return csr
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        torch.cumsum((csr.rowptr[1:] - csr.rowptr[:-1])[self._perm], 0, out=rowptr[1:])
        cols, vals = torch.empty_like(csr.cols), torch.empty_like(csr.vals)
        _lib.check(_lib.lib().tdr_csr_permute_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.cols), _lib.ptr(csr.vals), n, _lib.ptr(perm),
                                                  _lib.ptr(inv), _lib.ptr(rowptr), _lib.ptr(cols), _lib.ptr(vals), _lib.stream_ptr()),
                   "tdr_csr_permute_f32")
        return CSRAffinity(rowptr, cols, vals, row_offset=0, n_total=n)

    def _nn_for_exclusion(self):
        _, idx = self._csr.to_padded()
        return idx

    def on_affinity_computation_end(self):
        # plans and buffers of a previous fit (kept until clear_memory, which a fit that raised never reached) are sized
        # for THAT graph: never reuse them
        self._sched, self._grad_buf, self._sched_deferred, self._grad_ws, self._g = None, None, False, None, None
        self._pool = None
        super().on_affinity_computation_end()
        csr: CSRAffinity = self._relabel()
        self._csr_loop = csr
        L = _lib.lib()
        # the words the host looks at every check_interval iterations live in ONE buffer -- NaN flag, error words of the grouped layout and
        # of the schedule build, |grad| -- so that a check is one read instead of four (`_raise_if_nan` below)
        self._flagbuf = torch.zeros(4, dtype=torch.int32, device=csr.vals.device)
        if csr.vals.dtype == torch.float64:
            # float64 graph (float64 input): the epoch counters are float64 like the reference's (umap.py:215-234 in the
            # affinity's dtype) and the loop is the per-step float64 kernel, which reads the CSR as it is
            self.epochs_per_sample = torch.empty_like(csr.vals)
            self.epoch_of_next_sample = torch.empty_like(csr.vals)
            scratch = torch.zeros(2, dtype=torch.int64, device=csr.vals.device)
            _lib.check(L.tdr_umap_prepare_f64(_lib.ptr(csr.vals), csr.nnz, int(self.max_iter), _lib.ptr(self.epochs_per_sample),
                                              _lib.ptr(self.epoch_of_next_sample), _lib.ptr(scratch), _lib.stream_ptr()),
                       "tdr_umap_prepare_f64")
            self._loop_cols = csr.cols
            return
        eps_csr = torch.empty_like(csr.vals)
        nxt = torch.empty_like(csr.vals)
        scratch = torch.zeros(2, dtype=torch.int32, device=csr.vals.device)
        _lib.check(
            L.tdr_umap_prepare_f32(_lib.ptr(csr.vals), csr.nnz, int(self.max_iter), _lib.ptr(eps_csr), _lib.ptr(nxt),
                                   _lib.ptr(scratch), _lib.stream_ptr()),
            "tdr_umap_prepare_f32",
        )
        # loop layout: each row's edges by ascending epochs_per_sample (often-firing first).  The per-edge state of the
        # optimisation loop (epochs_per_sample, epoch_of_next_sample, _loop_cols) lives in this order; the affinity
        # graph itself keeps the reference's column order.
        self._loop_cols = torch.empty_like(csr.cols)
        self.epochs_per_sample = nxt  # reuse the buffer
        _lib.check(
            L.tdr_umap_sched_layout_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.cols), _lib.ptr(eps_csr), csr.n,
                                        _lib.ptr(self._loop_cols), _lib.ptr(self.epochs_per_sample), _lib.stream_ptr()),
            "tdr_umap_sched_layout_f32",
        )
        self.epoch_of_next_sample = self.epochs_per_sample.clone()  # umap.py:232
        if _opt("GROUPED") and _opt("SCHEDULED") and self.n_samples_in_ < 2**31 - 1 and self.n_components <= 32:
            dev, nnz = csr.vals.device, csr.nnz
            g = {"cols": torch.empty_like(csr.cols), "eps": torch.empty_like(csr.vals),
                 "rs": torch.empty(nnz, dtype=torch.uint8, device=dev), "order": torch.empty(nnz, dtype=torch.int32, device=dev),
                 "S": self._sched_slices(), "dirty": False}
            sc = self._sched_err = self._flagbuf[1:2]
            _lib.check(
                L.tdr_umap_sched_group_f32(_lib.ptr(csr.rowptr), _lib.ptr(self._loop_cols), _lib.ptr(self.epochs_per_sample), csr.n,
                                           self.n_samples_in_, g["S"], _lib.ptr(g["cols"]), _lib.ptr(g["eps"]), _lib.ptr(g["rs"]),
                                           _lib.ptr(g["order"]), _lib.ptr(sc), _lib.stream_ptr()),
                "tdr_umap_sched_group_f32",
            )
            g["next"] = g["eps"].clone()          # umap.py:232 in group order
            self._g = g

    # reference attributes (affinity_matcher.py:276-286) in their padded layout, materialised on request from the CSR
    @property
    def affinity_in_(self):
        if not hasattr(self, "_csr"):
            raise AttributeError("affinity_in_")
        return self._csr.to_padded()[0]

    @property
    def NN_indices_(self):
        if not hasattr(self, "_csr"):
            raise AttributeError("NN_indices_")
        return self._csr.to_padded()[1]

    def _fit_transform(self, X, y=None):
        nc_ok = range(1, 33) if _opt("SCHEDULED") else (2, 3)   # scheduled loop: exact kernels for 2 / 3, padded ones up to 32
        if self.n_components not in nc_ok:
            raise NotImplementedError(
                f"[torchdr_amd] UMAP: the HIP gradient kernels are built for n_components in 1..32 "
                f"(2 or 3 with SCHEDULED = False), got {self.n_components}."
            )
        return super()._fit_transform(X, y)    # NeighborEmbedding: un-permutes a loop that ran in cluster order

    def _sched_setup(self):
        """Static plan of the scheduled loop: list regions of the 64-row schedule blocks (one host read per fit)."""
        L, csr, dev = _lib.lib(), self._csr_loop, self.device_
        n_rows, nc = self.chunk_size_, self.n_components
        B = int(_opt("SCHED_BLOCK_ITERS"))
        S = self._sched_slices()
        g = getattr(self, "_g", None)
        if g is not None and g["S"] != S:
            raise RuntimeError("[torchdr_amd] UMAP: the number of L2 slices changed between the loop layout and the schedule plan.")
        rows_per_block = 16 if g is not None else 64
        n_blocks = (n_rows + rows_per_block - 1) // rows_per_block
        scratch = torch.empty(n_blocks, dtype=torch.int64, device=dev)
        blk_base = torch.empty(n_blocks + 1, dtype=torch.int64, device=dev)
        plan = L.tdr_umap_sched_plan_groups_f32 if g is not None else L.tdr_umap_sched_plan_f32
        _lib.check(plan(_lib.ptr(csr.rowptr), _lib.ptr(self.epochs_per_sample), n_rows, B,
                        _lib.ptr(scratch), _lib.ptr(blk_base), _lib.stream_ptr()),
                   "tdr_umap_sched_plan_f32")
        cap = int(blk_base[-1].item())
        # the joint launch relies on workgroups going round-robin over EIGHT XCDs that each cache one slice: only on the
        # whole device (256 CUs); on a partitioned one (e.g. one XCD per device) the slices go one launch at a time
        geom = int(_opt("SCHED_GEOM"))
        if torch.cuda.get_device_properties(dev).multi_processor_count < 256:
            geom &= 15
        self._sched = {
            "B": B, "S": S, "blk_base": blk_base, "t0": None, "n": 0,
            "list": torch.empty(cap + 64, dtype=torch.int32, device=dev),  # slack: idle lanes read entry 0 of a segment
            "hdr": torch.empty(2 * int(L.tdr_umap_sched_hdr_entries(n_rows, B, S)), dtype=torch.int32, device=dev),
            "err": self._flagbuf[2:3] if self.__dict__.get("_flagbuf") is not None else torch.zeros(1, dtype=torch.int32, device=dev),
            "geom": geom,
            # partial sums between the slice passes; the joint launch (geom & 16) keeps one plane per slice
            "acc": torch.empty(((S if geom & 16 else 1) * n_rows, 2 * nc), dtype=torch.float32, device=dev) if S > 1 else None,
        }
        return self._sched

    def _compute_gradients_scheduled(self, grad, neg, prof=False):
        L, csr = _lib.lib(), self._csr_loop
        sc = getattr(self, "_sched", None) or self._sched_setup()
        t = int(self.n_iter_)
        if sc["t0"] is None or not (sc["t0"] <= t < sc["t0"] + sc["n"]):
            n = max(1, min(sc["B"], int(self.max_iter) - t))
            g = getattr(self, "_g", None)
            if PROFILE is not None:
                eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                eb0.record()
            if g is not None:
                _lib.check(
                    L.tdr_umap_sched_build_groups_f32(_lib.ptr(csr.rowptr), _lib.ptr(g["cols"]), _lib.ptr(g["eps"]), _lib.ptr(g["rs"]),
                                                      _lib.ptr(g["next"]), self.chunk_size_, t, n, sc["S"], _lib.ptr(sc["blk_base"]),
                                                      _lib.ptr(sc["list"]), _lib.ptr(sc["hdr"]), _lib.ptr(sc["err"]),
                                                      int(_opt("SCHED_STAGE")), _lib.stream_ptr()),
                    "tdr_umap_sched_build_groups_f32",
                )
                g["dirty"] = True
            else:
                _lib.check(
                    L.tdr_umap_sched_build_f32(_lib.ptr(csr.rowptr), _lib.ptr(self._loop_cols), _lib.ptr(self.epochs_per_sample),
                                               _lib.ptr(self.epoch_of_next_sample), self.chunk_size_, self.n_samples_in_,
                                               t, n, sc["S"], _lib.ptr(sc["blk_base"]), _lib.ptr(sc["list"]),
                                               _lib.ptr(sc["hdr"]), _lib.ptr(sc["err"]), _lib.stream_ptr()),
                    "tdr_umap_sched_build_f32",
                )
            if PROFILE is not None:
                eb1.record()
                PROFILE.append(("build", eb0, eb1, n))
            sc["t0"], sc["n"] = t, n
        if prof:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        geom = sc["geom"]
        # joint launch + stock fused-SGD step: the gradient kernel leaves the per-slice planes and ONE kernel combines
        # them and steps the rows (`_sgd_kernel` below); anything that looks at the gradient in between keeps the form
        # with its own combine kernel
        self._sched_deferred = bool((geom & 16) and sc["S"] > 1 and self._fused_sgd and self._stock_step())
        if self._sched_deferred:
            geom |= 32
        if self._pool_negatives() and neg is None and self.embedding_.data_ptr() % 16 == 0:
            # POOL_FUSED_STEP: the launch also applies torch.optim.SGD's step (affinity_matcher.py:427) -- the stepped rows go to a
            # second embedding buffer and the two swap roles (`_sgd_kernel` below) -- when the step is the stock one without
            # momentum and nobody reads the gradient of this iteration (the reference inspects it every check_interval-th)
            fuse = (_opt("POOL_FUSED_STEP") and self._fused_sgd and float(self._sgd_momentum) == 0.0 and self._stock_step()
                    and t % max(int(self.check_interval), 1) != 0 and not PROFILE_KEEP_GRAD)
            if fuse:
                alt = self.__dict__.get("_Z_alt")
                if alt is None or alt.shape != self.embedding_.shape or alt.data_ptr() == self.embedding_.data_ptr():
                    alt = self._Z_alt = torch.empty_like(self.embedding_)
                    if self.world_size > 1:
                        alt.copy_(self.embedding_)      # the rows of other ranks arrive by the exchange; before the first one they must be valid
                _lib.check(
                    L.tdr_umap_pool_grad_step_f32(
                        _lib.ptr(self.embedding_), _lib.ptr(alt), self.n_components, self.n_samples_in_, self.chunk_start_, self.chunk_size_,
                        _lib.ptr(sc["list"]), _lib.ptr(sc["hdr"]), t - sc["t0"], float(self._a), float(self._b), t,
                        int(self.negative_sample_rate), int(self.n_negatives), self._neg_seed, float(self.early_exaggeration_coeff_),
                        float(self.repulsion_strength), float(self._eps), None, self._current_lr(), _lib.ptr(self._nan_flag),
                        int(_opt("POOL_GEOM")), _lib.stream_ptr(),
                    ),
                    "tdr_umap_pool_grad_step_f32",
                )
                self._pool_stepped = True
            else:
                _lib.check(
                    L.tdr_umap_pool_grad_f32(
                        _lib.ptr(self.embedding_), self.n_components, self.n_samples_in_, self.chunk_start_, self.chunk_size_,
                        _lib.ptr(sc["list"]), _lib.ptr(sc["hdr"]), t - sc["t0"], float(self._a), float(self._b), t,
                        int(self.negative_sample_rate), int(self.n_negatives), self._neg_seed, float(self.early_exaggeration_coeff_),
                        float(self.repulsion_strength), float(self._eps), _lib.ptr(grad), int(_opt("POOL_GEOM")), _lib.stream_ptr(),
                    ),
                    "tdr_umap_pool_grad_f32",
                )
            if prof:
                self._prof_pending = (ev0, ev1, csr.nnz)     # closed after the SGD step (_sgd_kernel): one whole iteration
            return
        _lib.check(
            L.tdr_umap_sched_grad_f32(
                _lib.ptr(self.embedding_), self.n_components, self.n_samples_in_, self.chunk_start_, self.chunk_size_,
                _lib.ptr(sc["list"]), _lib.ptr(sc["hdr"]), t - sc["t0"], sc["S"], float(self._a), float(self._b), t, int(self.negative_sample_rate), int(self.n_negatives),
                _lib.ptr(neg), self._neg_seed, float(self.early_exaggeration_coeff_), float(self.repulsion_strength),
                float(self._eps), _lib.ptr(grad), _lib.ptr(sc["acc"]), geom, _lib.stream_ptr(),
            ),
            "tdr_umap_sched_grad_f32",
        )
        if prof and self._sched_deferred:
            self._prof_pending = (ev0, ev1, csr.nnz)     # closed after the combine + SGD-step kernel (_sgd_kernel): one whole iteration
        elif prof:
            ev1.record()
            PROFILE.append(("grad", ev0, ev1, csr.nnz))

    def _stock_step(self) -> bool:
        """Nothing between the gradient evaluation and the optimizer step is overridden (hooks, step, gradient)."""
        from torchdr_amd.affinity_matcher import AffinityMatcher
        from torchdr_amd.neighbor_embedding.base import NeighborEmbedding

        cls = type(self)
        # the deferred combine + step writes the gradient only at the inspected iterations: anything that could look at
        # `_last_grad` in between (the step hooks, an own norm / convergence test, an own loop) keeps the undeferred form
        stock = (("on_training_step_end", NeighborEmbedding), ("_training_step", AffinityMatcher), ("_optimizer_step", AffinityMatcher),
                 ("_sgd_kernel", UMAP), ("_compute_gradients", UMAP), ("_compute_gradients_scheduled", UMAP),
                 ("on_training_step_start", NegativeSamplingNeighborEmbedding), ("_grad_norm", UMAP),
                 ("_converged", AffinityMatcher), ("_run_training_loop", UMAP))
        return all(getattr(cls, name) is getattr(owner, name) for name, owner in stock)

    def _sgd_kernel(self, Z, grad, chunk=False):
        if not getattr(self, "_sched_deferred", False):
            if self.__dict__.pop("_pool_stepped", False):
                # the gradient launch already wrote the stepped rows into the other buffer: the buffers swap roles (world_size > 1:
                # the exchange that follows fills in the other ranks' rows there)
                self.embedding_, self._Z_alt = self._Z_alt, self.embedding_
            else:
                super()._sgd_kernel(Z, grad, chunk=chunk)
            pend = self.__dict__.pop("_prof_pending", None)
            if pend is not None and PROFILE is not None:
                pend[1].record()
                PROFILE.append(("grad", pend[0], pend[1], pend[2]))
            return
        self._sched_deferred = False
        sc = self._sched
        mom = float(self._sgd_momentum)
        first = 0
        if mom != 0.0 and self._momentum_buf is None:
            self._momentum_buf, first = torch.empty_like(Z), 1
        # the gradient itself is read at the inspected iterations only (`_grad_norm`, every check_interval-th, and by the
        # hooks of subclasses, which never get here): elsewhere the pass does not write it
        want_grad = self.n_components not in (2, 3) or int(self.n_iter_) % max(int(self.check_interval), 1) == 0 or PROFILE_KEEP_GRAD
        _lib.check(
            _lib.lib().tdr_umap_sched_step_f32(_lib.ptr(sc["acc"]), sc["S"], self.n_components, self.chunk_size_,
                                               float(self.early_exaggeration_coeff_), float(self.repulsion_strength), _lib.ptr(grad if want_grad else None),
                                               _lib.ptr(Z), _lib.ptr(self._momentum_buf), self._current_lr(), mom, first,
                                               _lib.ptr(self._nan_flag), int(self.n_iter_), _lib.stream_ptr()),
            "tdr_umap_sched_step_f32",
        )
        pend = self.__dict__.pop("_prof_pending", None)
        if pend is not None and PROFILE is not None:
            pend[1].record()
            PROFILE.append(("grad", pend[0], pend[1], pend[2]))

    # ---- whole-loop runner ---------------------------------------------------------------------------------------------
    def _loop_runner_eligible(self) -> bool:
        from torchdr_amd.affinity_matcher import AffinityMatcher
        from torchdr_amd.neighbor_embedding.base import NegativeSamplingNeighborEmbedding, NeighborEmbedding

        if not (_opt("SCHEDULED") and _opt("LOOP_RUNNER")) or not self._fused_sgd or self.n_samples_in_ >= 2**31 - 1:
            return False
        if self._csr_loop.vals.dtype != torch.float32:
            return False
        if self.world_size > 1 and getattr(self, "_rccl_ctx", None) is None:
            return False
        if _opt("LOOP_RUNNER") == "auto" and self.world_size == 1:
            return False
        if self.neg_indices_ is not None or self._exclusion is not None or self.early_exaggeration_coeff_ > 1:
            return False
        cls = type(self)
        stock = (
            ("on_training_step_start", NegativeSamplingNeighborEmbedding), ("on_training_step_end", NeighborEmbedding),
            ("_training_step", AffinityMatcher), ("_optimizer_step", AffinityMatcher), ("_sgd_kernel", UMAP),
            ("_compute_gradients", UMAP), ("_compute_gradients_scheduled", UMAP), ("_grad_norm", UMAP),
        )
        return all(getattr(cls, name) is getattr(owner, name) for name, owner in stock)

    def _run_training_loop(self):
        fb = self.__dict__.get("_flagbuf")
        if fb is not None and fb.device == self._nan_flag.device:
            fb[0:1].copy_(self._nan_flag)
            self._nan_flag = fb[0:1]
        if not self._loop_runner_eligible():
            return super()._run_training_loop()
        import ctypes

        L, csr, dev = _lib.lib(), self._csr_loop, self.device_
        sc = getattr(self, "_sched", None) or self._sched_setup()
        T, ci, nc = int(self.max_iter), int(self.check_interval), self.n_components
        if getattr(self, "_grad_buf", None) is None:
            self._grad_buf = torch.empty((self.chunk_size_, nc), dtype=torch.float32, device=dev)
        mom = float(self._sgd_momentum)
        keep = {
            "lr": torch.tensor(self._lr_table[:T] + [0.0] * max(0, T - len(self._lr_table)), dtype=torch.float32, device=dev),
            "norm2": torch.zeros(T // max(ci, 1) + 2, dtype=torch.float32, device=dev),
            "scratch": torch.zeros(16, dtype=torch.int32, device=dev),
            "mom": torch.zeros_like(self._grad_buf) if mom != 0.0 else None,
            "snap": torch.empty_like(self._grad_buf),
        }
        d = _lib.UmapLoopDesc()
        d.Z, d.nc, d.n_total, d.row0, d.n_rows = _lib.ptr(self.embedding_), nc, self.n_samples_in_, self.chunk_start_, self.chunk_size_
        g = getattr(self, "_g", None)
        if g is not None:
            d.rowptr, d.cols, d.eps_per, d.next, d.rs = (_lib.ptr(csr.rowptr), _lib.ptr(g["cols"]), _lib.ptr(g["eps"]), _lib.ptr(g["next"]),
                                                         _lib.ptr(g["rs"]))
            g["dirty"] = True
        else:
            d.rowptr, d.cols, d.eps_per, d.next = (_lib.ptr(csr.rowptr), _lib.ptr(self._loop_cols), _lib.ptr(self.epochs_per_sample),
                                                   _lib.ptr(self.epoch_of_next_sample))
        d.blk_base, d.list, d.hdr, d.err = _lib.ptr(sc["blk_base"]), _lib.ptr(sc["list"]), _lib.ptr(sc["hdr"]), _lib.ptr(sc["err"])
        d.acc, d.grad, d.mom_buf = _lib.ptr(sc["acc"]), _lib.ptr(self._grad_buf), _lib.ptr(keep["mom"])
        d.a, d.b, d.neg_rate, d.n_negatives, d.seed = float(self._a), float(self._b), int(self.negative_sample_rate), int(self.n_negatives), self._neg_seed
        d.exag, d.rep, d.eps = float(self.early_exaggeration_coeff_), float(self.repulsion_strength), float(self._eps)
        d.n_slices, d.block_iters = sc["S"], min(sc["B"], max(ci, 1))
        d.lr_table, d.max_iter, d.momentum, d.first_iter, d.check_interval = _lib.ptr(keep["lr"]), T, mom, 0, ci
        d.norm2, d.snap, d.nan_flag, d.scratch = (_lib.ptr(keep["norm2"]), _lib.ptr(keep["snap"]), _lib.ptr(self._nan_flag),
                                                  _lib.ptr(keep["scratch"]))
        ctx = getattr(self, "_rccl_ctx", None)
        d.gather, d.gather_ctx = (ctx.gather_fn, ctx.handle) if ctx is not None else (None, None)
        capturable = ctx is not None and bool(getattr(ctx, "capturable", False))
        d.gather_capturable = 1 if capturable else 0
        d.geom = sc["geom"]
        d.pool = int(_opt("POOL_GEOM")) + 1 if (self._pool_negatives() and self.embedding_.data_ptr() % 16 == 0) else 0
        # POOL_FUSED_STEP: the step rides in the gradient launch (two embedding buffers; other ranks' rows of the second one are
        # delivered by the exchange of the iteration that writes it)
        if d.pool and _opt("POOL_FUSED_STEP") and mom == 0.0:
            keep["Z_alt"] = self.embedding_.clone()
            d.Z_alt = _lib.ptr(keep["Z_alt"])
        handle = ctypes.c_void_p()
        _lib.check(L.tdr_umap_loop_create(ctypes.byref(handle), ctypes.byref(d)), "tdr_umap_loop_create")
        # graphs cannot be captured on the legacy default stream: the loop runs on a side stream ordered after the
        # caller's stream, and the caller's stream waits for it at the end
        # windows that contain RCCL calls are enqueued as plain launches (captured collectives have never run on hardware here);
        # the peer exchange (and its loopback stand-in) keeps nothing on the host and is replayed with the window: at W = 8 a rank's
        # iteration is a few tens of microseconds of kernels and four launches of host work are as long (profiles/r06_rank_share*)
        self._loop_graph = bool(_opt("LOOP_GRAPH")) and (ctx is None or capturable)
        outer = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev) if self._loop_graph else outer
        side.wait_stream(outer)
        try:
            with torch.cuda.stream(side):
                self._loop_segments(L, handle, keep, T, ci, csr)
            sc["t0"], sc["n"] = None, 0
            self._last_grad, self._last_grad_is_chunk = self._grad_buf, True
        finally:
            outer.wait_stream(side)
            L.tdr_umap_loop_destroy(handle)

    def _loop_segments(self, L, handle, keep, T, ci, csr):
        """Whole windows (<= min(32, check_interval) iterations, so at most one inspected iteration each) run ahead of
        the host; after a window that holds an inspected iteration s (s % check_interval == 0) the host reads the NaN
        flag and |grad|(s) and, if the reference would have stopped at s (:343-349), restores the rows from the
        snapshot the step kernel took right after step s."""
        B = min(self._sched["B"], max(ci, 1))
        for w0 in range(0, T, B):
            n = min(B, T - w0)
            if LOOP_PROFILE is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _lib.check(L.tdr_umap_loop_run(handle, w0, n, 1 if self._loop_graph else 0, _lib.stream_ptr()), "tdr_umap_loop_run")
            if LOOP_PROFILE is not None:
                e1.record()
                LOOP_PROFILE.append((e0, e1, n, csr.nnz))
            self.n_iter_.fill_(w0 + n - 1)
            self._lr_pos = w0 + n
            s = -(-w0 // ci) * ci        # first inspected iteration >= w0
            if s < w0 + n:
                self._raise_if_nan()
                sq = keep["norm2"][s // ci].reshape(1).clone()
                if self.world_size > 1:
                    from torchdr_amd.parallel import allreduce_

                    allreduce_(sq)
                self._lr_pos = s + 1
                if self._converged(s, float(sq.sqrt().item())):
                    c0 = self.chunk_start_
                    self.embedding_[c0:c0 + self.chunk_size_].copy_(keep["snap"])
                    if self.world_size > 1:
                        from torchdr_amd.parallel import allgather_rows_

                        allgather_rows_(self.embedding_, c0, self.chunk_size_, self.world_size)
                    self.n_iter_.fill_(s)
                    return
                self._lr_pos = w0 + n

    def _raise_if_nan(self):
        fb = self.__dict__.get("_flagbuf")
        merged = (fb is not None and _opt("MERGED_CHECK") and self.world_size == 1 and getattr(self, "_fused_sgd", True) and getattr(self, "_rccl_ctx", None) is None
                  and self._nan_flag.data_ptr() == fb.data_ptr())
        if merged:
            # single process: ONE host read per check for the NaN flag, both schedule error words and the gradient norm (four round
            # trips before, each with the device idle behind it: ~1 ms of a 112 ms fit at 21 checks)
            import struct

            g = getattr(self, "_last_grad", None)
            have_norm = g is not None and not getattr(self, "_last_grad_is_chunk", False)
            if have_norm:
                fb.view(torch.float32)[3:4].copy_(g.norm(2).reshape(1))        # the value `_grad_norm` computes (affinity_matcher.py)
            nan_it, ge_v, se_v, nbits = fb.tolist()
            if have_norm:
                self._norm_cache = (g, struct.unpack("f", struct.pack("i", nbits))[0])
            if nan_it != 0:
                raise ValueError(f"[TorchDR] ERROR AffinityMatcher : NaNs in the embeddings at iter {nan_it - 1}.")
            if ge_v != 0 and self.__dict__.get("_sched_err") is not None:
                raise RuntimeError("[torchdr_amd] UMAP: a group of 16 rows holds more than 2^31 edges; set neighbor_embedding.umap.GROUPED = False.")
            if se_v != 0 and getattr(self, "_sched", None) is not None:
                raise RuntimeError("[torchdr_amd] UMAP: schedule build failed (list region overflow, a segment beyond 65535 "
                                   "entries or a list beyond 2^32 entries); set neighbor_embedding.umap.SCHEDULED = False.")
            return
        super()._raise_if_nan()
        sc = getattr(self, "_sched", None)
        ge = getattr(self, "_sched_err", None)
        if ge is not None and int(ge.item()) != 0:
            raise RuntimeError("[torchdr_amd] UMAP: a group of 16 rows holds more than 2^31 edges; set neighbor_embedding.umap.GROUPED = False.")
        if sc is not None and int(sc["err"].item()) != 0:
            raise RuntimeError("[torchdr_amd] UMAP: schedule build failed (list region overflow, a segment beyond 65535 "
                               "entries or a list beyond 2^32 entries); set neighbor_embedding.umap.SCHEDULED = False.")

    def _grad_norm(self) -> float:
        c = self.__dict__.pop("_norm_cache", None)
        if c is not None and c[0] is getattr(self, "_last_grad", None):
            return c[1]          # read together with the flags by `_raise_if_nan` of this check
        return super()._grad_norm()

    def _compute_gradients(self):
        csr: CSRAffinity = self._csr_loop
        if self.world_size > 1 or getattr(self, "_grad_buf", None) is None:
            self._grad_buf = torch.empty((self.chunk_size_, self.n_components), dtype=csr.vals.dtype,
                                         device=self.device_)
        grad = self._grad_buf
        neg = self._neg_ptr_tensor()
        if csr.vals.dtype == torch.float64:
            _lib.check(
                _lib.lib().tdr_umap_grad_f64(
                    _lib.ptr(self.embedding_), self.n_components, self.n_samples_in_, self.chunk_start_, self.chunk_size_,
                    _lib.ptr(csr.rowptr), _lib.ptr(self._loop_cols), _lib.ptr(self.epochs_per_sample),
                    _lib.ptr(self.epoch_of_next_sample), float(self._a), float(self._b), int(self.n_iter_),
                    int(self.negative_sample_rate), int(self.n_negatives), _lib.ptr(neg), self._neg_seed,
                    float(self.early_exaggeration_coeff_), float(self.repulsion_strength), float(self._eps), _lib.ptr(grad),
                    _lib.stream_ptr(),
                ),
                "tdr_umap_grad_f64",
            )
            return grad, True
        # sampled iterations sit between the inspected ones (every check_interval-th, where the step is a launch of its own)
        prof = PROFILE is not None and int(self.n_iter_) % PROFILE_EVERY == PROFILE_EVERY // 2
        if _opt("SCHEDULED") and self.n_samples_in_ < 2**31 - 1:
            self._compute_gradients_scheduled(grad, neg, prof)
            return grad, True
        if prof:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if getattr(self, "_grad_ws", None) is None:  # scratch of the L2-sliced negative phase (large N only)
            nbytes = _lib.lib().tdr_umap_grad_workspace_bytes(self.n_samples_in_, self.chunk_size_, self.n_components)
            self._grad_ws = torch.empty(max(nbytes, 8) // 4 + 1, dtype=torch.int32, device=self.device_)
            self._grad_ws_bytes = nbytes
        _lib.check(
            _lib.lib().tdr_umap_grad_f32(
                _lib.ptr(self.embedding_), self.n_components, self.n_samples_in_, self.chunk_start_,
                self.chunk_size_, _lib.ptr(csr.rowptr), _lib.ptr(self._loop_cols), _lib.ptr(self.epochs_per_sample),
                _lib.ptr(self.epoch_of_next_sample), float(self._a), float(self._b), int(self.n_iter_),
                int(self.negative_sample_rate), int(self.n_negatives), _lib.ptr(neg), self._neg_seed,
                float(self.early_exaggeration_coeff_), float(self.repulsion_strength), float(self._eps),
                _lib.ptr(grad), 0, _lib.ptr(self._grad_ws), self._grad_ws_bytes, _lib.stream_ptr(),
            ),
            "tdr_umap_grad_f32",
        )
        if prof:
            ev1.record()
            PROFILE.append(("grad", ev0, ev1, csr.nnz))
        return grad, True

    def clear_memory(self):
        super().clear_memory()
        self.__dict__.pop("_g", None)
        self.__dict__.pop("_pool", None)
        self.__dict__.pop("_Z_alt", None)
        self.__dict__.pop("_pool_stepped", None)
        self.__dict__.pop("_flagbuf", None)
        self.__dict__.pop("_norm_cache", None)
        self.__dict__.pop("_sched_err", None)
        for attr in ("_csr", "_csr_loop", "epochs_per_sample", "epoch_of_next_sample", "_exclusion", "_grad_buf", "_grad_ws", "_sched", "_loop_cols"):
            if hasattr(self, attr):
                delattr(self, attr)
