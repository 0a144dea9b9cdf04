"""GPU probe: the two schedule-build kernels on the N = 1M production graph (one window of 32 iterations each):
tdr_umap_sched_build_f32 (a lane owns an edge, CSR state) and tdr_umap_sched_build_ell_f32 (a lane owns a row, block-ELL state)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.conftest import gmm
from tests.test_umap_sched_gpu import Sched, SchedEll, layout, prepare
from torchdr_amd import _lib
from torchdr_amd.affinity import UMAPAffinity
from torchdr_amd.distance.base import ClusterIndex, PackedPoints

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
X = gmm(n, 128, 2.0).cuda()
csr = UMAPAffinity(n_neighbors=30, max_iter=100)(X, return_csr=True)
ci = ClusterIndex(PackedPoints(X))
perm, inv = ci.perm, ci.inv
WIN = int(os.environ.get("DEG_WINDOW", "0"))
if WIN:   # refine the cluster-sorted order: inside windows of WIN consecutive positions, rows by descending degree
    deg0 = (csr.rowptr[1:] - csr.rowptr[:-1])[perm.long()]
    key = (torch.arange(n, device="cuda") // WIN) * 8192 + (8191 - deg0.clamp(max=8191))
    order = torch.sort(key, stable=True).indices
    perm = perm[order].contiguous()
    inv = torch.empty_like(perm)
    inv[perm.long()] = torch.arange(n, dtype=torch.int32, device="cuda")
rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
torch.cumsum((csr.rowptr[1:] - csr.rowptr[:-1])[perm.long()], 0, out=rowptr[1:])
cols, vals = torch.empty_like(csr.cols), torch.empty_like(csr.vals)
L = _lib.lib()
_lib.check(L.tdr_csr_permute_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.cols), _lib.ptr(csr.vals), n, _lib.ptr(perm), _lib.ptr(inv),
                                 _lib.ptr(rowptr), _lib.ptr(cols), _lib.ptr(vals), _lib.stream_ptr()), "permute")
del X, ci
eps_per, _ = prepare(vals, 1000)
cols, eps_per = layout(rowptr, cols, eps_per)
deg = rowptr[1:] - rowptr[:-1]
wb = deg[: (n // 64) * 64].view(-1, 64).max(1).values
print(json.dumps({"nnz": int(rowptr[-1]), "mean_deg": float(deg.float().mean()), "max_deg": int(deg.max()),
                  "mean_block_width": float(wb.float().mean()), "max_block_width": int(wb.max())}), flush=True)


def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return sum(ts[:4]) / 4


a = Sched(rowptr, cols, eps_per, n, 32, S)
nxt = eps_per.clone()
for t0 in (0, 32, 64):
    a.build(nxt, t0, 32)
snap = nxt.clone()


def old():
    nxt.copy_(snap)
    a.build(nxt, 96, 32)


b = SchedEll(rowptr, cols, eps_per, n, 32, S)
b._pack(snap)
ell_snap = b.ell_next.clone()


def new():
    b.ell_next.copy_(ell_snap)
    _lib.check(L.tdr_umap_sched_build_ell_f32(_lib.ptr(b.ell_base), _lib.ptr(b.ell_row), _lib.ptr(b.ell_cols), _lib.ptr(b.ell_eps),
                                              _lib.ptr(b.ell_next), _lib.ptr(b.ell_mask), n, n, 96, 32, S, _lib.ptr(b.blk_base),
                                              _lib.ptr(b.list), _lib.ptr(b.hdr), _lib.ptr(b.err), _lib.stream_ptr()), "ell")


print(json.dumps({"edge_per_lane_ms": timed(old) - timed(lambda: nxt.copy_(snap)),
                  "row_per_lane_ms": timed(new) - timed(lambda: b.ell_next.copy_(ell_snap))}), flush=True)
