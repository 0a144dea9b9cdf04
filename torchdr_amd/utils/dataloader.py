"""DataLoader inputs (reference ``distance/base.py:121-157``, ``distance/faiss.py`` *_from_dataloader,
``affinity_matcher.py:218-234``).

The reference streams batches into a Faiss index because the point set may not fit next to its N x N intermediates.
On an MI355X the points themselves fit (288 GB of HBM = 500 M points at D = 128) and nothing of size N^2 is ever
formed, so the batches are streamed straight into ONE resident device tensor (the host never holds the full set)
and the regular exact kernels run on it."""

import numpy as np
import torch
from torch.utils.data import DataLoader


def is_dataloader(x) -> bool:
    return isinstance(x, DataLoader)


def _first(batch):
    if isinstance(batch, (list, tuple)):
        batch = batch[0]
    if isinstance(batch, np.ndarray):
        batch = torch.from_numpy(batch)
    if not isinstance(batch, torch.Tensor) or batch.dim() != 2:
        raise ValueError("[TorchDR] DataLoader batches must be 2-D tensors (or tuples whose first item is one).")
    return batch


def dataloader_metadata(dl: DataLoader):
    """(n_samples, n_features, dtype, device) from the dataset length and the first batch."""
    try:
        n = len(dl.dataset)
    except TypeError:
        n = None
    for batch in dl:
        b = _first(batch)
        return n, b.shape[1], b.dtype, b.device
    raise ValueError(
        "[TorchDR] DataLoader is empty, cannot determine metadata. Ensure DataLoader yields at least one batch."
    )


def _iterates_in_order(sampler) -> bool:
    from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler

    if isinstance(sampler, BatchSampler):
        return _iterates_in_order(sampler.sampler)
    if isinstance(sampler, RandomSampler):
        return False
    if isinstance(sampler, SequentialSampler):
        return True
    return not getattr(sampler, "shuffle", False)


def check_dataloader_order(dl: DataLoader):
    """Neighbour indices refer to the order the loader yields its rows in, so that order has to be the dataset's
    (reference ``distance/faiss.py:60-110``: a shuffling sampler is refused, an unknown one draws a warning)."""
    sampler = getattr(dl, "sampler", None)
    if sampler is None:
        import warnings

        warnings.warn("[TorchDR] Could not verify DataLoader has shuffle=False. Ensure deterministic iteration for correct "
                      "k-NN results.")
        return
    if not _iterates_in_order(sampler):
        raise ValueError(
            "[TorchDR] DataLoader must have shuffle=False for deterministic iteration. Current sampler: "
            f"{type(sampler).__name__}. k-NN indices will be incorrect with shuffled data."
        )


_METADATA = "_tdr_metadata"      # kept on the loader object itself: no global table that outlives it


def get_dataloader_metadata(dl: DataLoader):
    """{'n_samples', 'n_features', 'dtype', 'device'} of a loader that has been through `materialize_dataloader`, else None
    (reference ``distance/faiss.py:27-41``)."""
    return getattr(dl, _METADATA, None)


def materialize_dataloader(dl: DataLoader, device=None) -> torch.Tensor:
    """All batches, in iteration order, as one (n_samples, n_features) tensor on ``device`` (default: the current GPU
    when there is one).  Integer batches are cast to float32 like tensor inputs."""
    check_dataloader_order(dl)
    if device is None or device == "auto":
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    n, d, dtype, src_device = dataloader_metadata(dl)
    if not dtype.is_floating_point:
        dtype = torch.float32

    def remember(rows):
        try:
            setattr(dl, _METADATA, {"n_samples": int(rows), "n_features": int(d), "dtype": dtype, "device": src_device})
        except Exception:   # a loader class that refuses new attributes: nothing cached
            pass

    if n is None:  # iterable dataset: length unknown until exhausted
        parts = [_first(b).to(device=device, dtype=dtype) for b in dl]
        remember(sum(p.shape[0] for p in parts))
        return torch.cat(parts)
    out = torch.empty((n, d), dtype=dtype, device=device)
    pos = 0
    for batch in dl:
        b = _first(batch)
        if b.shape[1] != d:
            raise ValueError(f"[TorchDR] DataLoader batches disagree on the number of features ({b.shape[1]} vs {d}).")
        m = b.shape[0]
        if pos + m > n:
            raise ValueError(f"[TorchDR] DataLoader yielded more than len(dataset) = {n} samples.")
        out[pos:pos + m].copy_(b, non_blocking=True)
        pos += m
    remember(pos)
    return out if pos == n else out[:pos]      # drop_last=True loaders yield fewer rows


def stream_dataloader_packed(dl: DataLoader, device=None, metric: str = "sqeuclidean"):
    """``pairwise_distances(dataloader, k=...)``: the batches go straight into the search's own layout -- each batch is
    copied into its rows of the resident block and the MFMA tile images / norms of the tiles it completes are packed at
    once (``PackedPoints.from_batches``), instead of materialising the block and packing it in a second pass.  Returns
    ``(X, packed)``; ``packed`` is None where the search does not use the fp32 tile images (D > 256, manhattan /
    sqhyperbolic, non-float32 batches, iterable datasets of unknown length): the block is then materialised only."""
    from torchdr_amd.distance.base import PackedPoints

    check_dataloader_order(dl)
    if device is None or device == "auto":
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    n, d, dtype, src_device = dataloader_metadata(dl)
    if not dtype.is_floating_point:
        dtype = torch.float32
    if (n is None or d > 256 or dtype != torch.float32 or metric not in ("sqeuclidean", "euclidean", "angular")
            or torch.device(device).type != "cuda"):
        return materialize_dataloader(dl, device), None
    packed = PackedPoints.from_batches((_first(b) for b in dl), n, d, torch.device(device))
    try:
        setattr(dl, _METADATA, {"n_samples": int(packed.n), "n_features": int(d), "dtype": dtype, "device": src_device})
    except Exception:
        pass
    return packed.X, packed
