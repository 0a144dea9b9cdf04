"""GPU probe: time variants of the UMAP gradient kernel on the N=1M graph (what bounds it?)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.conftest import gmm
from torchdr_amd import _lib
if os.environ.get("TDR_LIB"):
    _lib.LIB_PATH = os.environ["TDR_LIB"]
from torchdr_amd.affinity import UMAPAffinity

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X = gmm(n, 128, 2.0).cuda()
csr = UMAPAffinity(n_neighbors=30, max_iter=100)(X, return_csr=True)
del X
L = _lib.lib()
nnz = csr.nnz
eps_per = torch.empty(nnz, device="cuda"); nxt0 = torch.empty(nnz, device="cuda")
scratch = torch.zeros(2, dtype=torch.int32, device="cuda")
_lib.check(L.tdr_umap_prepare_f32(_lib.ptr(csr.vals), nnz, 1000, _lib.ptr(eps_per), _lib.ptr(nxt0), _lib.ptr(scratch), _lib.stream_ptr()), "prep")
Z = (torch.randn(n, 2, device="cuda") * 5).contiguous()
grad = torch.empty((n, 2), device="cuda")
print(json.dumps({"n": n, "nnz": nnz, "mean_deg": nnz / n}))

ws = torch.empty(n * 4 + 16, dtype=torch.int32, device="cuda")


def run(name, neg_rate, n_neg, neg_inj, iters=30, t0=100, slices=0):
    nxt = nxt0.clone()
    # advance counters to a steady-state iteration
    ts = []
    for it in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.tdr_umap_grad_f32(_lib.ptr(Z), 2, n, 0, n, _lib.ptr(csr.rowptr), _lib.ptr(csr.cols), _lib.ptr(eps_per),
                                       _lib.ptr(nxt), 1.577, 0.895, t0 + it, neg_rate, n_neg, _lib.ptr(neg_inj), 1234, 1.0, 1.0,
                                       1e-3, _lib.ptr(grad), slices, _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), "grad")
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts = sorted(ts)[: max(1, len(ts) // 2)]
    print(json.dumps({"variant": name, "ms": sum(ts) / len(ts)}), flush=True)

sl = int(os.environ.get("SL", "0"))
run(f"slices={sl} pos_geom={os.environ.get('TDR_UMAP_POS_GEOM','0')} neg_geom={os.environ.get('TDR_UMAP_NEG_GEOM','0')}", 5, 150, None, slices=sl)
run("no_negatives", 0, 150, None)
