"""Scoped overrides of the package's tuning / behaviour switches.

The switches live as module attributes (``distance.base.PRUNE_MODE``, ``neighbor_embedding.umap.SCHEDULED`` ...): they are the
process-wide defaults.  Code that wants another value for ONE call -- a test, a benchmark, a caller that must not disturb
other threads -- uses :func:`options`:

    with torchdr_amd.config.options(PRUNE_MODE="force", RELABEL=False):
        Z = torchdr_amd.UMAP(...).fit_transform(X)

The override is carried by a ``contextvars.ContextVar``: it is visible to the code the ``with`` block calls (in this thread /
task only), nests, and is undone on exit even when the block raises -- setting and restoring module attributes by hand is
neither re-entrant nor exception-safe.  The modules read every switch through :func:`get`.
"""

import contextvars
from contextlib import contextmanager

_OVERRIDES = contextvars.ContextVar("torchdr_amd_options", default=None)

# switch -> module that defines its default
SWITCHES = {
    "SCREEN_MODE": "torchdr_amd.distance.base", "PRUNE_MODE": "torchdr_amd.distance.base",
    "PILOT_CONCURRENT": "torchdr_amd.distance.base", "WIDE_SCAN": "torchdr_amd.distance.base",
    "TILE_BOUNDS": "torchdr_amd.distance.base", "REFINE_INDEX": "torchdr_amd.distance.base", "PRUNED_LISTS": "torchdr_amd.distance.base", "FLAT_SCAN": "torchdr_amd.distance.base", "FLAT_TWO_TERMS": "torchdr_amd.distance.base", "FLAT_FORCE_TERMS": "torchdr_amd.distance.base", "ASSIGN16": "torchdr_amd.distance.base", "FLAT_WS_LIMIT": "torchdr_amd.distance.base",
    "SCHEDULED": "torchdr_amd.neighbor_embedding.umap", "RELABEL": "torchdr_amd.neighbor_embedding.umap",
    "LOOP_RUNNER": "torchdr_amd.neighbor_embedding.umap", "LOOP_GRAPH": "torchdr_amd.neighbor_embedding.umap",
    "SCHED_GEOM": "torchdr_amd.neighbor_embedding.umap", "MERGED_CHECK": "torchdr_amd.neighbor_embedding.umap", "SCHED_SLICES": "torchdr_amd.neighbor_embedding.umap",
    "SCHED_BLOCK_ITERS": "torchdr_amd.neighbor_embedding.umap", "GROUPED": "torchdr_amd.neighbor_embedding.umap",
    "SCHED_STAGE": "torchdr_amd.neighbor_embedding.umap", "NEGATIVES": "torchdr_amd.neighbor_embedding.umap",
    "POOL_GEOM": "torchdr_amd.neighbor_embedding.umap", "POOL_FUSED_STEP": "torchdr_amd.neighbor_embedding.umap",
    "PCA_EIGH": "torchdr_amd.affinity_matcher", "PCA_PREFETCH": "torchdr_amd.affinity_matcher",
    "RCCL_CONTEXT": "torchdr_amd.neighbor_embedding.base", "PERM_NEGATIVES": "torchdr_amd.neighbor_embedding.base",
    "PEER_EXCHANGE": "torchdr_amd.neighbor_embedding.base",
}


def get(name: str, module_globals: dict):
    """Value of switch `name`: the innermost active override, else the defining module's attribute."""
    ov = _OVERRIDES.get()
    if ov is not None and name in ov:
        return ov[name]
    return module_globals[name]


@contextmanager
def options(**overrides):
    unknown = [k for k in overrides if k not in SWITCHES]
    if unknown:
        raise KeyError(f"[torchdr_amd] unknown option(s) {unknown}; known: {sorted(SWITCHES)}")
    cur = _OVERRIDES.get()
    token = _OVERRIDES.set({**(cur or {}), **overrides})
    try:
        yield
    finally:
        _OVERRIDES.reset(token)
