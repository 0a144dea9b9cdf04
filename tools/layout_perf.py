"""GPU probe: the per-row sorts of the fit outside the loop on a synthetic graph of the headline's shape (N rows, ~43 edges per row):
tdr_umap_sched_layout_f32 (rows by (period, column)), timed alone with HIP events, plus a plain copy of the same bytes as the floor.

    python tools/layout_perf.py [N]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from torchdr_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
real = len(sys.argv) > 2 and sys.argv[2] == "real"      # the headline's own graph (UMAPAffinity of the BASELINE mixture) instead of a synthetic one
L = _lib.lib()
gen = torch.Generator().manual_seed(0)
if real:
    from tests.conftest import gmm
    from torchdr_amd.affinity import UMAPAffinity

    csr = UMAPAffinity(n_neighbors=30)(gmm(n, 128, 2.0).cuda(), return_csr=True)
    rowptr, cols = csr.rowptr, csr.cols
    eps = (csr.vals.max() / (csr.vals + 1e-3)).contiguous()
    nnz = csr.nnz
    deg = (rowptr[1:] - rowptr[:-1]).cpu()
else:
    deg = torch.randint(30, 57, (n,), generator=gen)
    deg[::97] = torch.randint(65, 300, (deg[::97].numel(),), generator=gen)
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = deg.cumsum(0)
    nnz = int(rowptr[-1])
    cols = torch.randint(0, n, (nnz,), generator=gen, dtype=torch.int32).cuda()
    eps = (1.0 / (torch.rand(nnz, generator=gen) ** 3 + 1e-3)).cuda()
    rowptr = rowptr.cuda()
co, eo = torch.empty_like(cols), torch.empty_like(eps)


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def layout():
    _lib.check(L.tdr_umap_sched_layout_f32(_lib.ptr(rowptr), _lib.ptr(cols), _lib.ptr(eps), n, _lib.ptr(co), _lib.ptr(eo), _lib.stream_ptr()), "layout")


def copy():
    co.copy_(cols)
    eo.copy_(eps)


out = {"n": n, "nnz": nnz, "real_graph": real, "rows_over_64": float((deg > 64).float().mean()), "rows_over_128": float((deg > 128).float().mean()),
       "rows_over_256": float((deg > 256).float().mean()), "max_deg": int(deg.max()), "layout_ms": round(timed(layout), 3), "copy_ms": round(timed(copy), 3)}
# check: every row sorted by (period, column)
layout()
rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (rowptr[1:] - rowptr[:-1]))
key = eo.double() * 0  # placeholder to keep the check cheap: periods ascending inside a row
same = rows[1:] == rows[:-1]
out["sorted"] = bool(((eo[1:] >= eo[:-1]) | ~same).all())
print(json.dumps(out))
