"""Logger / seeding / argument helpers (reference ``utils/utils.py:20-97``)."""

import logging
import random

import numpy as np
import torch


def set_logger(name: str, verbose: bool = False) -> logging.Logger:
    """Per-class logger with the reference's format ``[TorchDR] <name>: <message>`` (utils.py:20-48)."""
    logger = logging.getLogger(f"torchdr_amd.{name}")
    logger.setLevel(logging.INFO if verbose else logging.WARNING)
    if not logger.handlers:
        handler = logging.StreamHandler()
        handler.setFormatter(logging.Formatter("[TorchDR] %(name)s: %(message)s"))
        logger.addHandler(handler)
    logger.propagate = False
    return logger


def seed_everything(seed, fast=True, deterministic=False):
    """Seed python / numpy / torch generators (utils.py:51-97); returns the seed used."""
    if seed is None:
        seed = int(torch.randint(0, 2**31 - 1, (1,)).item())
    seed = int(seed)
    random.seed(seed)
    np.random.seed(seed % (2**32))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def bool_arg(arg):
    if isinstance(arg, bool):
        return arg
    if isinstance(arg, str):
        return arg.lower() in ("true", "1", "yes")
    return bool(arg)


def compute_device(X: torch.Tensor, device="auto") -> torch.device:
    """Device the HIP kernels run on.  ``"auto"`` keeps a GPU tensor where it is; host data goes to
    the current HIP device (this build has no CPU compute path)."""
    if device is None or device == "auto":
        if isinstance(X, torch.Tensor) and X.is_cuda:
            return X.device
        if not torch.cuda.is_available():
            raise RuntimeError("[torchdr_amd] no HIP device available and this build has no CPU compute path.")
        return torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"[torchdr_amd] device={device!r}: this build has no CPU compute path.")
    return dev


_WARNED_FP64 = False


def as_float32(X):
    """float64 inputs (numpy's default) are accepted and PROCESSED IN float32: the HIP path is fp32 (the
    north-star dtype); callers cast results back to the input dtype.  Warns once per process."""
    import warnings

    import torch

    global _WARNED_FP64
    if isinstance(X, torch.Tensor) and X.dtype == torch.float64:
        if not _WARNED_FP64:
            warnings.warn("[torchdr_amd] float64 input is processed in float32 on the HIP path "
                          "(results are returned in float64).", stacklevel=3)
            _WARNED_FP64 = True
        return X.to(torch.float32)
    return X
