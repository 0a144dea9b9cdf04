"""Multi-GPU exchange steps of the hot path (one process per GPU, RCCL over xGMI through
``torch.distributed``; ``gloo`` works for the CPU-side tests of the routing logic).

Call sites being replaced:
  C2  affinity_matcher.py:395-413  zero-padded all-reduce of per-rank rows  -> all-gather of rows
  C3  affinity_matcher.py:425      all-reduce of the full gradient          -> all-reduce (RCCL)
  C4/C5 utils/sparse.py:259-309    edge routing for the symmetrisation      -> all-to-all-v with
        INTEGER (int32) index payload (the reference ships indices as fp32, exact only below 2^24).
"""

from typing import Tuple

import torch
import torch.distributed as dist

from torchdr_amd.distributed import DistributedContext, chunk_bounds


# Measurement of ONE rank's share of a W-rank fit on one GPU (utils/emulation.py, tools/rank_share.py, bench.py --emulate-rank): the
# process group is torch's "fake" backend (rank r of W, collectives that move nothing) and the two exchanges that CARRY data are
# served by this object -- the transposed edges of the symmetrisation from the other ranks' graphs (computed beforehand), the
# per-iteration row exchange as a local copy of the same bytes.  None in every real run.
EMULATION = None


def _host_staged() -> bool:
    """gloo has no device collectives for every op used here; stage through host memory then
    (CPU-side tests and single-GPU multi-process debugging).  RCCL ("nccl") works on device buffers."""
    return dist.get_backend() == "gloo"


def allreduce_(t: torch.Tensor):
    if _host_staged() and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_max_(t: torch.Tensor):
    if _host_staged() and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        t.copy_(h)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def broadcast_(t: torch.Tensor, src: int = 0):
    if _host_staged() and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
        return t
    dist.broadcast(t, src=src)
    return t


def _all_to_all_single(out, inp, out_splits=None, in_splits=None):
    if _host_staged() and inp.is_cuda:
        ho, hi = out.cpu(), inp.cpu()
        dist.all_to_all_single(ho, hi, output_split_sizes=out_splits, input_split_sizes=in_splits)
        out.copy_(ho)
        return
    dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits)


def allgather_rows(local_rows: torch.Tensor, n_total: int, world_size: int) -> torch.Tensor:
    """Concatenate every rank's row chunk (chunks differ by at most one row)."""
    sizes = [chunk_bounds(n_total, r, world_size) for r in range(world_size)]
    max_rows = max(e - s for s, e in sizes)
    nc = local_rows.shape[1]
    pad = local_rows
    if local_rows.shape[0] < max_rows:
        pad = torch.cat([local_rows, local_rows.new_zeros(max_rows - local_rows.shape[0], nc)])
    out = local_rows.new_empty((world_size * max_rows, nc))
    if _host_staged() and pad.is_cuda:
        ho = out.cpu()
        dist.all_gather_into_tensor(ho, pad.contiguous().cpu())
        out.copy_(ho)
    else:
        dist.all_gather_into_tensor(out, pad.contiguous())
    if all(e - s == max_rows for s, e in sizes):
        return out
    parts = [out[r * max_rows: r * max_rows + (e - s)] for r, (s, e) in enumerate(sizes)]
    return torch.cat(parts)


def allgather_rows_(full: torch.Tensor, chunk_start: int, chunk_size: int, world_size: int) -> torch.Tensor:
    """In-place form for the per-iteration exchange: ``full[chunk_start : chunk_start + chunk_size]`` holds this rank's
    updated rows; on return every rank's rows are in ``full``.  With equal chunks the rank's send buffer IS its slot of
    the receive buffer (the in-place all-gather RCCL supports: no staging copy, no scratch allocation per iteration);
    uneven chunks or host-staged backends fall back to :func:`allgather_rows`."""
    if EMULATION is not None:
        return EMULATION.allgather_rows_(full)
    n = full.shape[0]
    if n == chunk_size * world_size and full.is_contiguous() and not (_host_staged() and full.is_cuda):
        dist.all_gather_into_tensor(full, full[chunk_start: chunk_start + chunk_size])
        return full
    full.copy_(allgather_rows(full[chunk_start: chunk_start + chunk_size], n, world_size))
    return full


def route_edges(values: torch.Tensor, indices: torch.Tensor, chunk_start: int, n_total: int, world_size: int,
                rank: int):
    """Split this rank's edges (i -> j, v) by the owner of j.  Returns per-destination lists of
    (src_global int32, dst_global int32, v fp32) for destinations != rank (pure tensor logic; no comm)."""
    n, k = values.shape
    src = (torch.arange(n, device=values.device, dtype=torch.int64) + chunk_start).repeat_interleave(k)
    dst = indices.reshape(-1).to(torch.int64)
    v = values.reshape(-1)
    owner = DistributedContext.get_rank_for_indices(dst, n_total, world_size)
    order = torch.argsort(owner, stable=True)
    owner_s, src_s, dst_s, v_s = owner[order], src[order], dst[order], v[order]
    counts = torch.bincount(owner_s, minlength=world_size)
    out = []
    offs = 0
    for r, c in enumerate(counts.tolist()):
        sl = slice(offs, offs + c)
        offs += c
        if r == rank:
            out.append(None)
        else:
            out.append((src_s[sl].to(torch.int32), dst_s[sl].to(torch.int32), v_s[sl].contiguous()))
    return out


def exchange_transposed_edges(values, indices, chunk_start, n_total, world_size) -> Tuple[torch.Tensor, ...]:
    """All-to-all-v of the edges whose transpose lives on another rank (reference sparse.py:259-309).
    Returns (ext_row int32 local, ext_col int32 global, ext_val in the dtype of ``values``) for ``symmetrize_to_csr``."""
    if EMULATION is not None:
        return EMULATION.transposed_edges(values, indices, chunk_start, n_total, world_size)
    rank = dist.get_rank()
    dev = values.device
    routed = route_edges(values, indices, chunk_start, n_total, world_size, rank)
    send_counts = torch.tensor([0 if p is None else p[0].numel() for p in routed], dtype=torch.int64, device=dev)
    recv_counts = torch.empty_like(send_counts)
    _all_to_all_single(recv_counts, send_counts)
    sc, rc = send_counts.tolist(), recv_counts.tolist()

    def cat(idx, dtype):
        parts = [p[idx] for p in routed if p is not None and p[idx].numel() > 0]
        return torch.cat(parts) if parts else torch.empty(0, dtype=dtype, device=dev)

    send_src, send_dst, send_v = cat(0, torch.int32), cat(1, torch.int32), cat(2, values.dtype)
    total = int(sum(rc))
    recv_src = torch.empty(total, dtype=torch.int32, device=dev)
    recv_dst = torch.empty(total, dtype=torch.int32, device=dev)
    recv_v = torch.empty(total, dtype=values.dtype, device=dev)      # float32, or float64 on the float64 path
    _all_to_all_single(recv_src, send_src, rc, sc)
    _all_to_all_single(recv_dst, send_dst, rc, sc)
    _all_to_all_single(recv_v, send_v, rc, sc)
    # received edge (src -> dst) with dst owned here: it is entry (dst, src) of P^T
    return (recv_dst - chunk_start).to(torch.int32), recv_src, recv_v


def exchange_rows_to_owners(row_ids, vals, idx, n_total, world_size, chunk_start, chunk_size):
    """kNN rows computed for arbitrary global rows (``row_ids`` int32, ``vals`` (m, k) fp32, ``idx`` (m, k) int32) ->
    the rows of this rank's chunk, in row order.  All-to-all-v by owner rank (contiguous chunks, the first
    ``n_total % world_size`` ranks hold one extra row)."""
    dev = vals.device
    k = vals.shape[1]
    owner = DistributedContext.get_rank_for_indices(row_ids.long(), n_total, world_size)
    order = torch.argsort(owner, stable=True)
    send_counts = torch.bincount(owner, minlength=world_size).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    _all_to_all_single(recv_counts, send_counts)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    total = int(sum(rc))
    s_ids, s_v, s_i = row_ids[order].contiguous(), vals[order].contiguous(), idx[order].contiguous()
    r_ids = torch.empty(total, dtype=torch.int32, device=dev)
    r_v = torch.empty((total, k), dtype=torch.float32, device=dev)
    r_i = torch.empty((total, k), dtype=torch.int32, device=dev)
    _all_to_all_single(r_ids, s_ids, rc, sc)
    _all_to_all_single(r_v, s_v, rc, sc)
    _all_to_all_single(r_i, s_i, rc, sc)
    if total != chunk_size:
        raise RuntimeError(f"[torchdr_amd] kNN row exchange: received {total} rows for a chunk of {chunk_size}.")
    out_v = torch.empty((chunk_size, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((chunk_size, k), dtype=torch.int32, device=dev)
    local = (r_ids.long() - chunk_start)
    out_v[local] = r_v
    out_i[local] = r_i
    return out_v, out_i


def gather_row_shards(X_local: torch.Tensor) -> torch.Tensor:
    """Row-sharded input (north star: "the point set shards by rows ... with RCCL all-gather"): every rank passes ITS rows,
    in rank order, and gets the full (N, D) block back -- one all-gather of the shards over xGMI, after which the
    search runs exactly as on a replicated input (queries = own chunk, database = all N).  Shards must follow the
    reference's chunk rule (``distributed/__init__.py:209-219``: contiguous, the first ``N mod W`` ranks hold one extra
    row) so that the rank's shard IS its chunk of every later stage."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if X_local.dim() != 2:
        raise ValueError("[TorchDR] ERROR : a row shard must be a 2-D block.")
    staged = _host_staged() and X_local.is_cuda
    dev = torch.device("cpu") if staged or not X_local.is_cuda else X_local.device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    sizes[rank] = X_local.shape[0]
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM)
    sizes = sizes.tolist()
    n = int(sum(sizes))
    for r, m in enumerate(sizes):
        s, e = chunk_bounds(n, r, world)
        if m != e - s:
            raise ValueError(
                f"[TorchDR] ERROR : row shards must follow the chunk rule of DistributedContext.compute_chunk_bounds "
                f"(rank {r} holds {m} rows, expected {e - s} of {n})."
            )
    max_rows = max(sizes)
    loc = X_local.contiguous()
    if staged:
        loc = loc.cpu()
    if loc.shape[0] < max_rows:
        loc = torch.cat([loc, loc.new_zeros((max_rows - loc.shape[0], loc.shape[1]))])
    out = loc.new_empty((world * max_rows, loc.shape[1]))
    dist.all_gather_into_tensor(out, loc)
    if any(m != max_rows for m in sizes):
        out = torch.cat([out[r * max_rows: r * max_rows + m] for r, m in enumerate(sizes)])
    return out.to(X_local.device)


class RcclContext:
    """``tdr_ctx_*``: an RCCL communicator owned by the C library, whose collectives are enqueued on the caller's stream
    (capturable into the HIP graphs of the UMAP loop object).  Rank 0 draws the unique id and the bytes travel through
    the already initialised ``torch.distributed`` group.  ``create`` returns None instead of raising when RCCL cannot be
    opened or the self-check against ``torch.distributed`` fails: callers then keep the Python-level collectives."""

    def __init__(self, handle, n_total):
        import ctypes

        from torchdr_amd import _lib

        self.handle = handle
        self.n_total = n_total
        self.gather_fn = ctypes.cast(_lib.lib().tdr_ctx_allgather_rows, ctypes.c_void_p)

    @staticmethod
    def rccl_path() -> bytes:
        import os

        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return cand.encode() if os.path.exists(cand) else b""

    @classmethod
    def create(cls, n_total: int, device, rank=None, world=None, broadcast=None):
        import ctypes

        from torchdr_amd import _lib

        L = _lib.lib()
        rank = dist.get_rank() if rank is None else rank
        world = dist.get_world_size() if world is None else world
        path = cls.rccl_path()
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            if L.tdr_ctx_unique_id(path, buf) != 0:
                uid.fill_(255)   # "unavailable" marker, broadcast like a real id so that every rank gives up together
            else:
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if world > 1:
            if broadcast is not None:
                uid = broadcast(uid)
            else:
                t = uid.to(device) if dist.get_backend() == "nccl" else uid
                dist.broadcast(t, src=0)
                uid = t.cpu()
        if bool((uid == 255).all()):
            return None
        handle = ctypes.c_void_p()
        raw = bytes(uid.numpy().tobytes())
        with torch.cuda.device(device):
            rc = L.tdr_ctx_create(ctypes.byref(handle), rank, world, raw, path, n_total)
        ctx = cls(handle, n_total) if rc == 0 else None
        ok = ctx is not None and ctx._self_check(device, rank, world)
        if world > 1 and broadcast is None:
            # every rank keeps the context or none does (a rank that fell back alone would wait in a torch.distributed
            # collective the others never enter)
            flag = torch.tensor([1.0 if ok else 0.0], device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item() > 0)
        if not ok:
            if ctx is not None:
                ctx.destroy()
            return None
        return ctx

    # ONE communicator per process and device, shared by every fit: ncclCommInitRank costs tens of milliseconds (bootstrap,
    # topology, channel set-up) -- per fit it would be most of a row-sharded fit_transform at 8 GPUs
    _shared = {}

    @staticmethod
    def _group_token():
        """Identity of the current default process group: a communicator belongs to the group it was bootstrapped in -- after
        destroy_process_group() / init_process_group() (tests, elastic restarts) the same (device, world, rank) may name
        other peers.  A counter, not id(): the object of the group a token was issued for is kept alive until another group
        takes its place, so a new group can never be handed the address -- and with it the communicator or the IPC mappings
        -- of a dead one."""
        global _GROUP_SEEN
        try:
            pg = dist.distributed_c10d._get_default_group()
        except Exception:
            return None
        if _GROUP_SEEN is None or _GROUP_SEEN[0] is not pg:
            _GROUP_SEEN = (pg, (_GROUP_SEEN[1] + 1) if _GROUP_SEEN is not None else 1)
        return _GROUP_SEEN[1]

    @classmethod
    def shared(cls, n_total: int, device):
        """The process's communicator on `device` for the CURRENT process group, re-targeted at an embedding of `n_total` rows;
        created (collectively: every rank reaches its first call in the same fit) on first use.  None when RCCL is
        unavailable -- a failed creation is not remembered (the next fit tries again, collectively).  Fits of one process
        share the communicator and its row count: they are expected one after the other (one fit at a time per process)."""
        from torchdr_amd import _lib

        token = cls._group_token()
        for k in [k for k in cls._shared if k[3] != token]:     # communicators of a process group that no longer exists
            ctx = cls._shared.pop(k)
            if ctx is not None:
                ctx.destroy()
        key = (torch.device(device).index, dist.get_world_size(), dist.get_rank(), token)
        ctx = cls._shared.get(key)
        if ctx is None:
            ctx = cls.create(n_total, device)
            if ctx is None:
                return None
            if not cls._shared:
                import atexit

                atexit.register(cls.destroy_shared)
            cls._shared[key] = ctx
        if ctx.n_total != n_total:
            _lib.check(_lib.lib().tdr_ctx_set_rows(ctx.handle, n_total), "tdr_ctx_set_rows")
            ctx.n_total = n_total
        return ctx

    @classmethod
    def destroy_shared(cls):
        for ctx in cls._shared.values():
            if ctx is not None:
                ctx.destroy()
        cls._shared.clear()

    def _self_check(self, device, rank, world) -> bool:
        """An all-gather of 3 columns through the context must place every rank's rows where chunk_bounds says."""
        from torchdr_amd import _lib

        Z = torch.full((self.n_total, 3), -1.0, dtype=torch.float32, device=device)
        s, e = chunk_bounds(self.n_total, rank, world)
        Z[s:e] = float(rank + 1)
        if _lib.lib().tdr_ctx_allgather_rows(self.handle, _lib.ptr(Z), 3, _lib.stream_ptr()) != 0:
            return False
        want = torch.empty(self.n_total, dtype=torch.float32, device=device)
        for r in range(world):
            a, b = chunk_bounds(self.n_total, r, world)
            want[a:b] = float(r + 1)
        return bool((Z == want[:, None]).all())

    def allgather_rows_(self, Z: torch.Tensor):
        from torchdr_amd import _lib

        if Z.shape[0] != self.n_total:      # another fit re-targeted the shared communicator in between
            _lib.check(_lib.lib().tdr_ctx_set_rows(self.handle, Z.shape[0]), "tdr_ctx_set_rows")
            self.n_total = Z.shape[0]

        _lib.check(_lib.lib().tdr_ctx_allgather_rows(self.handle, _lib.ptr(Z), Z.shape[1], _lib.stream_ptr()),
                   "tdr_ctx_allgather_rows")
        return Z

    def allreduce_(self, t: torch.Tensor):
        from torchdr_amd import _lib

        _lib.check(_lib.lib().tdr_ctx_allreduce_f32(self.handle, _lib.ptr(t), t.numel(), _lib.stream_ptr()), "tdr_ctx_allreduce_f32")
        return t

    def destroy(self):
        from torchdr_amd import _lib

        if self.handle is not None:
            _lib.lib().tdr_ctx_destroy(self.handle)
            self.handle = None


_GROUP_SEEN = None      # (default process group object, its token): see RcclContext._group_token


class PeerExchange:
    """``tdr_peerx_*`` (csrc/tdr_peerx.hip): the per-iteration all-gather of the rows every rank stepped as direct peer writes
    over xGMI -- each rank stores its chunk into a staging block of every peer (all links at once), raises a generation flag,
    waits for the flags raised at it and copies the staged chunks into its embedding -- instead of a ring collective.  Same
    interface as :class:`RcclContext` for the row exchange (``gather_fn`` / ``handle`` for the C loop object,
    ``allgather_rows_``).  ``create`` returns None when the peers cannot be mapped (HIP IPC) or the stress self-check fails on
    ANY rank: callers then keep RCCL / torch.distributed."""

    capturable = True       # the exchange's generation lives in device memory: windows that contain it replay as HIP graphs
    _shared = {}
    _retired = set()        # process-group tokens whose exchange failed once: later fits of that group use RCCL / torch.distributed
    SELF_CHECK_ROUNDS = 12
    WAIT_LIMIT = None       # spins of the bounded flag wait (None = the library's default, a few seconds); tdr_peerx_set_wait_limit

    def __init__(self, handle, n_total, capacity):
        import ctypes

        from torchdr_amd import _lib

        self.handle, self.n_total, self.capacity = handle, n_total, capacity
        self.gather_fn = ctypes.cast(_lib.lib().tdr_peerx_allgather_rows, ctypes.c_void_p)

    @classmethod
    def create(cls, n_total: int, nc: int, device):
        import ctypes

        from torchdr_amd import _lib

        L = _lib.lib()
        rank, world = dist.get_rank(), dist.get_world_size()
        if world < 2 or world > 16:
            return None
        capacity = int(n_total) * max(int(nc), 3)
        handle = ctypes.c_void_p()
        ok = True
        with torch.cuda.device(device):
            ok = L.tdr_peerx_create(ctypes.byref(handle), rank, world, capacity) == 0
            mine = ctypes.create_string_buffer(128)
            ok = ok and L.tdr_peerx_handles(handle, mine) == 0
        everyone = [None] * world
        dist.all_gather_object(everyone, (bool(ok), bytes(mine.raw)))
        if not all(o for o, _ in everyone):
            if handle.value:
                L.tdr_peerx_destroy(handle)
            return None
        blob = b"".join(h for _, h in everyone)
        with torch.cuda.device(device):
            opened = L.tdr_peerx_open(handle, blob) == 0
        ctx = cls(handle, n_total, capacity)
        L.tdr_peerx_set_rows(handle, n_total)
        if cls.WAIT_LIMIT is not None:
            L.tdr_peerx_set_wait_limit(handle, int(cls.WAIT_LIMIT))
        good = opened and ctx._self_check(device, rank, world)
        flag = torch.tensor([1.0 if good else 0.0], device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)     # every rank keeps the exchange or none does
        if not bool(flag.item() > 0):
            ctx.destroy()
            return None
        return ctx

    def _self_check(self, device, rank, world) -> bool:
        """Several exchanges of CHANGING patterns through the production kernels (1 / 2 / 3 columns, uneven chunks as they
        come), every row compared on every rank after every round: what a stale cache line or a lost flag would break."""
        from torchdr_amd import _lib

        L = _lib.lib()
        for rnd in range(self.SELF_CHECK_ROUNDS):
            nc = 1 + rnd % 3
            Z = torch.full((self.n_total, nc), -1.0, dtype=torch.float32, device=device)
            s, e = chunk_bounds(self.n_total, rank, world)
            Z[s:e] = float(1000 * rnd + rank + 1)
            with torch.cuda.device(device):
                if L.tdr_peerx_allgather_rows(self.handle, _lib.ptr(Z), nc, _lib.stream_ptr()) != 0:
                    return False
            want = torch.empty(self.n_total, dtype=torch.float32, device=device)
            for r in range(world):
                a, b = chunk_bounds(self.n_total, r, world)
                want[a:b] = float(1000 * rnd + r + 1)
            if not bool((Z == want[:, None]).all()):
                return False
        return L.tdr_peerx_error(self.handle) == 0

    @classmethod
    def shared(cls, n_total: int, nc: int, device):
        """The process's exchange on `device` for the current process group (created collectively on first use, re-created when
        an embedding no longer fits its stages); None when unavailable."""
        from torchdr_amd import _lib

        token = RcclContext._group_token()
        for k in [k for k in cls._shared if k[3] != token]:
            cls._shared.pop(k).destroy()
        if token in cls._retired:
            return None
        key = (torch.device(device).index, dist.get_world_size(), dist.get_rank(), token)
        ctx = cls._shared.get(key)
        if ctx is not None and int(n_total) * int(nc) > ctx.capacity:
            cls._shared.pop(key).destroy()
            ctx = None
        if ctx is None:
            ctx = cls.create(n_total, nc, device)
            if ctx is None:
                return None
            if not cls._shared:
                import atexit

                atexit.register(cls.destroy_shared)
            cls._shared[key] = ctx
        if ctx.n_total != n_total:
            _lib.check(_lib.lib().tdr_peerx_set_rows(ctx.handle, n_total), "tdr_peerx_set_rows")
            ctx.n_total = n_total
        return ctx

    @classmethod
    def destroy_shared(cls):
        for ctx in cls._shared.values():
            ctx.destroy()
        cls._shared.clear()

    @classmethod
    def retire_shared(cls):
        """After a wait of the exchange ran into its limit (every rank agreed on it, affinity_matcher._raise_if_nan): the
        contexts are destroyed on all ranks -- stages, flags, the device-side error word and the host-side generation counters
        go with them, so nothing of the failed exchange can leak into a later fit -- and the current process group falls back
        to the RCCL all-gather / torch.distributed for the rest of its life."""
        cls.destroy_shared()
        token = RcclContext._group_token()
        if token is not None:
            cls._retired.add(token)

    def allgather_rows_(self, Z: torch.Tensor):
        from torchdr_amd import _lib

        if Z.shape[0] != self.n_total:
            _lib.check(_lib.lib().tdr_peerx_set_rows(self.handle, Z.shape[0]), "tdr_peerx_set_rows")
            self.n_total = Z.shape[0]
        _lib.check(_lib.lib().tdr_peerx_allgather_rows(self.handle, _lib.ptr(Z), Z.shape[1], _lib.stream_ptr()), "tdr_peerx_allgather_rows")
        return Z

    def failed(self) -> bool:
        from torchdr_amd import _lib

        return _lib.lib().tdr_peerx_error(self.handle) != 0

    def destroy(self):
        from torchdr_amd import _lib

        if self.handle is not None:
            _lib.lib().tdr_peerx_destroy(self.handle)
            self.handle = None
