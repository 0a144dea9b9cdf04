"""One rank's share of a W-rank row-sharded fit, measured on ONE GPU.

No node with more than one MI355X has been available to this build; what a rank of a W-rank fit does is nevertheless a
function of (X, rank, W) except for two exchanges.  ``EmulatedRank`` runs that rank alone:

* the process group is torch's ``fake`` backend -- ``dist.get_rank()`` / ``get_world_size()`` answer (rank, W), collectives move
  nothing: the seed broadcast, the pilots' MAX vote, the NaN-flag and gradient-norm reductions keep the rank's own value;
* ``exchange_transposed_edges`` (the all-to-all-v of the symmetrisation, reference utils/sparse.py:259-309) is served from the
  directed graphs of the OTHER ranks, which ``collect()`` computes beforehand by running each rank's search + bandwidth stage in
  turn (untimed) -- the rank receives exactly the edges it would receive;
* the per-iteration row exchange (reference affinity_matcher.py:395-413) is ``tdr_emulx_allgather_rows``: the rank's chunk written
  W - 1 times and the N - chunk rows of the peers copied into the embedding -- the bytes of the exchange without the links (the
  peers' rows stay where the initialisation put them).  It plugs into the C loop object exactly as the RCCL / peer-write contexts
  do, so the host cost per iteration is the production one.

What the emulation cannot show is link time and skew between ranks; the table built from it adds the transfer
as bytes / link bandwidth and says so.
"""

import ctypes

import torch
import torch.distributed as dist

from torchdr_amd import _lib
from torchdr_amd.distributed import DistributedContext, chunk_bounds


class _Collected(Exception):
    pass


class _LoopbackExchange:
    """``RcclContext`` / ``PeerExchange`` look-alike around ``tdr_emulx_*``."""

    capturable = True

    def __init__(self, rank, world, n_total, nc, device):
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().tdr_emulx_create(ctypes.byref(self.handle), rank, world, n_total, nc), "tdr_emulx_create")
        self.gather_fn = ctypes.cast(_lib.lib().tdr_emulx_allgather_rows, ctypes.c_void_p)
        self.n_total, self.nc = n_total, nc

    def allgather_rows_(self, Z):
        _lib.check(_lib.lib().tdr_emulx_allgather_rows(self.handle, _lib.ptr(Z), Z.shape[1], _lib.stream_ptr()), "tdr_emulx_allgather_rows")
        return Z

    def destroy(self):
        if self.handle is not None:
            _lib.lib().tdr_emulx_destroy(self.handle)
            self.handle = None


class EmulatedRank:
    def __init__(self, world: int):
        self.world = int(world)
        self.rank = None
        self.graphs = {}        # rank -> (values (n_r, k), indices (n_r, k), chunk_start)
        self.collecting = False
        self._ctx = None
        self.exchange_bytes = None

    # ---- process group ------------------------------------------------------------------------------------------------
    def enter(self, rank: int):
        from torch.testing._internal.distributed.fake_pg import FakeStore

        from torchdr_amd import parallel

        if dist.is_initialized():
            dist.destroy_process_group()
        dist.init_process_group(backend="fake", rank=int(rank), world_size=self.world, store=FakeStore())
        self.rank = int(rank)
        parallel.EMULATION = self

    def leave(self):
        from torchdr_amd import parallel

        parallel.EMULATION = None
        if self._ctx is not None:
            self._ctx.destroy()
            self._ctx = None
        if dist.is_initialized():
            dist.destroy_process_group()

    # ---- the two exchanges that carry data ------------------------------------------------------------------------------
    def transposed_edges(self, values, indices, chunk_start, n_total, world_size):
        if self.collecting:
            self.graphs[self.rank] = (values.detach().clone(), indices.detach().clone(), int(chunk_start))
            raise _Collected()
        from torchdr_amd.parallel import route_edges

        # what the rank itself does in a real run: split ITS edges by owner (the send side; nobody receives them here) ...
        route_edges(values, indices, chunk_start, n_total, world_size, self.rank)
        # ... and what it receives: the edges of the other ranks that end in its rows, prepared by `prepare()` outside the timed run
        src, dst, val = self._received[self.rank]
        return (dst - chunk_start).to(torch.int32), src, val.to(values.dtype)

    def prepare(self, rank, n_total):
        """The edges rank `rank` receives (in the order an all-to-all-v delivers them: by source rank), from the collected graphs."""
        from torchdr_amd.parallel import route_edges

        src, dst, val = [], [], []
        dev = None
        for r in range(self.world):
            if r == rank:
                continue
            v, i, c0 = self.graphs[r]
            dev = v.device
            part = route_edges(v, i, c0, n_total, self.world, r)[rank]
            if part is not None:
                src.append(part[0]); dst.append(part[1]); val.append(part[2])
        if src:
            got = (torch.cat(src), torch.cat(dst), torch.cat(val))
        else:
            e = torch.empty(0, dtype=torch.int32, device=dev)
            got = (e, e.clone(), torch.empty(0, dtype=torch.float32, device=dev))
        self.__dict__.setdefault("_received", {})[rank] = got
        self.edge_exchange_bytes = int(got[0].numel()) * 12
        return got

    def exchange(self, n_total, nc, device):
        if self._ctx is None or self._ctx.n_total != n_total or self._ctx.nc != nc:
            if self._ctx is not None:
                self._ctx.destroy()
            self._ctx = _LoopbackExchange(self.rank, self.world, n_total, nc, device)
        c0, c1 = chunk_bounds(n_total, self.rank, self.world)
        self.exchange_bytes = {"sent_per_iteration": (c1 - c0) * nc * 4 * (self.world - 1), "received_per_iteration": (n_total - (c1 - c0)) * nc * 4}
        return self._ctx

    def allgather_rows_(self, full):
        return self.exchange(full.shape[0], full.shape[1], full.device).allgather_rows_(full)

    # ---- driver -----------------------------------------------------------------------------------------------------------
    def collect(self, make_affinity, X, skip=None):
        """Run every rank's search + bandwidth stage in turn (untimed) and keep its directed graph; ``make_affinity()`` returns a
        fresh affinity object (constructed INSIDE the emulated group, so that it picks the rank up)."""
        self.collecting = True
        try:
            for r in range(self.world):
                if skip is not None and r == skip and r in self.graphs:
                    continue
                self.enter(r)
                aff = make_affinity()
                try:
                    aff(X, return_indices=True, return_csr=True)
                except _Collected:
                    pass
                else:
                    raise RuntimeError("[torchdr_amd] emulation: the affinity did not reach the edge exchange")
                del aff
        finally:
            self.collecting = False
