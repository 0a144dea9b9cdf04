"""Row-sharding contract for multi-GPU runs (one process per GPU, RCCL over xGMI).

Mirror of the reference's ``torchdr/distributed/__init__.py:115-319`` (``DistributedContext``,
``is_distributed``/``get_rank``/``get_world_size``): contiguous row chunks, the first
``N mod W`` ranks get one extra row (``:209-219``), inverse map (``:251-267``).

Differences by design (MI355X-first):
  * the process group is NOT initialised at import time; ``init_from_env()`` does it when
    ``LOCAL_RANK`` is set (backend "nccl" == RCCL on ROCm; "gloo" for the CPU-side tests of
    the host logic);
  * there is no Faiss config -- every rank runs the same exact HIP search on its chunk.
"""

import os
from typing import Tuple

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_rank() -> int:
    return dist.get_rank() if is_distributed() else 0


def get_world_size() -> int:
    return dist.get_world_size() if is_distributed() else 1


def init_from_env(backend: str = None) -> bool:
    """Initialise ``torch.distributed`` from torchrun's environment (no-op when absent)."""
    if is_distributed():
        return True
    if "LOCAL_RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return False
    local_rank = int(os.environ["LOCAL_RANK"])
    if backend is None:
        # TDR_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses duplicate devices) -- debugging / CI only,
        # the collectives are then staged through host memory (parallel._host_staged)
        backend = os.environ.get("TDR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())  # more ranks than GPUs only in the gloo debugging mode
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend)
    return True


class DistributedContext:
    """Rank / world-size holder plus the row-chunk arithmetic.

    ``force_enable`` marks the context initialised without a process group (tests set
    ``rank`` / ``world_size`` by hand, as the reference's own tests do).
    """

    def __init__(self, force_enable: bool = False):
        initialised = is_distributed()
        self.is_initialized = initialised or force_enable
        if initialised:
            self.rank = dist.get_rank()
            self.world_size = dist.get_world_size()
            self.local_rank = int(os.environ.get("LOCAL_RANK", 0))
            if torch.cuda.is_available():
                self.local_rank %= torch.cuda.device_count()
                torch.cuda.set_device(self.local_rank)
        else:
            self.rank = 0
            self.world_size = 1
            self.local_rank = 0

    def compute_chunk_bounds(self, n_samples: int) -> Tuple[int, int]:
        """[start, end) of this rank's rows (reference ``distributed/__init__.py:183-219``)."""
        return chunk_bounds(n_samples, self.rank, self.world_size)

    @staticmethod
    def get_rank_for_indices(indices: torch.Tensor, n_samples: int, world_size: int) -> torch.Tensor:
        """Owner rank of each global row index (reference ``:221-267``)."""
        base = n_samples // world_size
        rem = n_samples % world_size
        split = rem * (base + 1)
        if base == 0:
            return torch.clamp(indices, 0, world_size - 1)
        ranks = torch.where(indices < split, indices // (base + 1), rem + (indices - split) // base)
        return torch.clamp(ranks, 0, world_size - 1)

    def get_faiss_config(self, base_config=None):
        """``FaissConfig`` for this rank's GPU (reference ``distributed/__init__.py:269-309``): the caller's settings with
        the device replaced by the local rank."""
        from torchdr_amd.distance import FaissConfig

        if base_config is None:
            return FaissConfig(device=self.local_rank)
        return FaissConfig(temp_memory=base_config.temp_memory, device=self.local_rank, index_type=base_config.index_type,
                           nprobe=base_config.nprobe, nlist=base_config.nlist, **base_config.faiss_kwargs)

    def __repr__(self):
        if self.is_initialized:
            return (
                f"DistributedContext(rank={self.rank}, world_size={self.world_size}, "
                f"local_rank={self.local_rank})"
            )
        return "DistributedContext(not initialized)"


def chunk_bounds(n_samples: int, rank: int, world_size: int) -> Tuple[int, int]:
    base = n_samples // world_size
    rem = n_samples % world_size
    if rank < rem:
        start = rank * (base + 1)
        return start, start + base + 1
    start = rank * base + rem
    return start, start + base
