"""ms per launch of the dense TSNE repulsion (tdr_tsne_repulsion_f32, N^2 pairs) -- python tools/tsne_rep_perf.py [lib.so]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from torchdr_amd import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
L = _lib.lib()
SPLIT = os.environ.get("SPLIT", "1") != "0"
out = {}
for n, nc in ((5_000, 2), (20_000, 2), (50_000, 2), (100_000, 2), (200_000, 2), (100_000, 3)):
    Z = (torch.randn(n, nc, generator=torch.Generator().manual_seed(0)) * 10).cuda().contiguous()
    F = torch.empty((n, nc), device="cuda")
    S = torch.zeros(1, dtype=torch.float64, device="cuda")

    nb = int(L.tdr_tsne_repulsion_workspace_bytes(n, n, nc)) if (SPLIT and hasattr(L, "tdr_tsne_repulsion_workspace_bytes")) else 0
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")

    def run():
        S.zero_()
        if nb:
            _lib.check(L.tdr_tsne_repulsion_split_f32(_lib.ptr(Z), nc, n, 0, n, _lib.ptr(F), _lib.ptr(S), _lib.ptr(ws), nb, _lib.stream_ptr()), "rep")
        else:
            _lib.check(L.tdr_tsne_repulsion_f32(_lib.ptr(Z), nc, n, 0, n, _lib.ptr(F), _lib.ptr(S), _lib.stream_ptr()), "rep")

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    out[f"n={n},nc={nc}"] = {"ms": round(ms, 3), "pairs_per_s": n * n / ms * 1e3, "S": float(S), "F_abs_sum": float(F.abs().sum())}
print(json.dumps(out))
