"""Time of the PCA initialisation's eigensolvers on one GPU: python tools/eigh_perf.py  -> one JSON line per D."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from torchdr_amd import _lib

L = _lib.lib()
for d in (50, 64, 128, 256):
    gen = torch.Generator().manual_seed(d)
    X = torch.randn(4000, d, generator=gen, dtype=torch.float64)
    G = (X.T @ X).cuda()
    ev = torch.empty(d, dtype=torch.float64, device="cuda")
    V = torch.empty((d, d), dtype=torch.float64, device="cuda")
    ws = torch.empty(2 * d * d, dtype=torch.float64, device="cuda")
    out = {"d": d}
    for name, fn in (("top2", lambda: L.tdr_eigh_top_f64(_lib.ptr(G), d, 2, _lib.ptr(ev), _lib.ptr(V), _lib.ptr(ws), _lib.stream_ptr())),
                     ("top4", lambda: L.tdr_eigh_top_f64(_lib.ptr(G), d, 4, _lib.ptr(ev), _lib.ptr(V), _lib.ptr(ws), _lib.stream_ptr())),
                     ("jacobi", lambda: L.tdr_eigh_jacobi_f64(_lib.ptr(G), d, _lib.ptr(ev), _lib.ptr(V), _lib.ptr(ws), _lib.stream_ptr()))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name + "_ms"] = round(e0.elapsed_time(e1) / 3, 4)
    print(json.dumps(out), flush=True)
