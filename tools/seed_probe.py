"""GPU probe: the adaptive farthest-point seeding on the bench generator: seeds chosen and kernel time for several c_min."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import gmm
from torchdr_amd import _lib
from torchdr_amd.distance.base import PackedPoints, dense_packed

L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X = gmm(n, 128, 2.0).cuda()
S = 8192
idx = torch.empty(S, dtype=torch.int32, device="cuda")
_lib.check(L.tdr_cluster_sample_i32(n, S, 20240917, _lib.ptr(idx), _lib.stream_ptr()), "sample")
Xs = X[idx.long()].contiguous()
print("blobs in sample:", int(torch.unique(idx % 1000).numel()))
Ps = PackedPoints(Xs)
D2 = dense_packed(Ps, Ps, "sqeuclidean", False)
for c_min in (500, 999, 1000, 1001, 1200):
    seeds = torch.empty(2048, dtype=torch.int32, device="cuda")
    ns = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(L.tdr_cluster_maxmin_adaptive_f32(_lib.ptr(D2), D2.stride(0), S, c_min, 2048, 0.25, _lib.ptr(seeds), _lib.ptr(ns), _lib.stream_ptr()), "mm")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = int(ns)
    sd = seeds[:max(k, 1002)].long()
    # max-min distances of the seeds around the 1000th, recomputed
    dm = D2[sd][:, sd]
    deltas = [float(dm[i, :i].min()) for i in (998, 999, 1000, 1001)] if sd.numel() > 1001 else []
    print(c_min, "->", k, "seeds", round(dt * 1e3, 2), "ms", [round(x, 1) for x in deltas])
