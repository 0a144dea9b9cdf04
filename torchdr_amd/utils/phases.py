"""Per-phase timing of a fit (HIP events on the launch stream): ``bench.py --gpus N`` reports where a row-sharded
``fit_transform`` spends its time (shard gather, kNN, exchanges, symmetrisation, init, loop).  Off by default -- a
``with phase(...)`` costs one attribute read then."""

from contextlib import contextmanager

import torch

# a list while recording: (name, start_event, end_event); None = off
RECORD = None


@contextmanager
def phase(name: str):
    rec = RECORD
    if rec is None or not torch.cuda.is_available():
        yield
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    try:
        yield
    finally:
        e1.record()
        rec.append((name, e0, e1))


def start():
    global RECORD
    RECORD = []


def stop():
    """Stop recording; returns {name: milliseconds} summed over the recorded intervals (synchronises)."""
    global RECORD
    rec, RECORD = RECORD, None
    out = {}
    if rec:
        torch.cuda.synchronize()
        for name, e0, e1 in rec:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
    return out
