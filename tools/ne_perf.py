"""GPU probe: LargeVis / TSNE sparse-gradient kernel variants at N=1M (what bounds it?)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torchdr_amd import _lib
L = _lib.lib()
n, k = 1_000_000, 15
g = torch.Generator(device="cuda").manual_seed(0)
# synthetic kNN-like graph: neighbours within +-500 of the row (local) -- enough for timing
nn = ((torch.arange(n, device="cuda")[:, None] + torch.randint(1, 500, (n, k), device="cuda", generator=g)) % n).to(torch.int32).contiguous()
P = torch.rand(n, k, device="cuda", generator=g) / n
Z = (torch.randn(n, 2, device="cuda", generator=g) * 5).contiguous()
grad = torch.zeros(n, 2, device="cuda")
from torchdr_amd.neighbor_embedding.base import build_transposed_graph
TG = (None, None, None)
def run(name, kk, n_neg):
    ts = []
    for it in range(20):
        grad.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.tdr_ne_grad_f32(_lib.ptr(Z), 2, n, 0, n, _lib.ptr(nn), _lib.ptr(P), kk, _lib.ptr(TG[0]), _lib.ptr(TG[1]), _lib.ptr(TG[2]), 0, 1.0, 2.0 / n, n_neg, None, 123, it,
                                     _lib.ptr(grad), _lib.stream_ptr()), "ne")
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts = sorted(ts)[:10]
    print(json.dumps({"variant": name, "ms": sum(ts) / len(ts)}), flush=True)
run("push k15+5neg", 15, 5)
run("push k15 only", 15, 0)
TG = build_transposed_graph(P, nn, 0, n, 1)
run("pull k15+5neg", 15, 5)
run("pull k15 only", 15, 0)
nn1 = nn[:, :1].contiguous(); P1 = P[:, :1].contiguous()
def run1(name, n_neg):
    ts = []
    for it in range(20):
        grad.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.tdr_ne_grad_f32(_lib.ptr(Z), 2, n, 0, n, _lib.ptr(nn1), _lib.ptr(P1), 1, None, None, None, 0, 1.0, 2.0 / n, n_neg, None, 123, it,
                                     _lib.ptr(grad), _lib.stream_ptr()), "ne")
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts = sorted(ts)[:10]
    print(json.dumps({"variant": name, "ms": sum(ts) / len(ts)}), flush=True)
run1("k1+5neg", 5)
run1("k1 only", 0)
