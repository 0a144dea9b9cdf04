"""GPU lab for the threshold scan (csrc/tdr_knn_flat.hip): whole-search times with the threshold scan on / off on the three
unpruned workloads of bench.py (mixture with pruning off, structureless, uniform k = 15), then the scan kernel alone per
(terms, shape) against realistic thresholds.   python tools/knn_flat_lab.py [n] > gpurun_out/knn_flat_lab.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tests.conftest import gmm
from torchdr_amd import _lib, config
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances


def timed(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    d = 128
    res = {"n": n, "d": d}
    torch.manual_seed(42)
    sets = {"mixture_s2_k30": (gmm(n, d, 2.0), 30), "structureless_k30": (gmm(n, d, 0.0), 30), "uniform_k15": (torch.randn(n, d), 15)}
    keep = None
    only_scan = len(sys.argv) > 2 and sys.argv[2] == "scan"
    if only_scan:
        X = sets["mixture_s2_k30"][0].cuda()
        with config.options(PRUNE_MODE="0", FLAT_SCAN=True):
            C, I = pairwise_distances(X, metric="sqeuclidean", k=30, exclude_diag=True, return_indices=True)
        keep = (X, C)
        sets = {}
    for name, (Xc, k) in sets.items():
        X = Xc.cuda()
        row = {}
        outs = {}
        for flat in (True, False):
            with config.options(PRUNE_MODE="0", FLAT_SCAN=flat):
                t, (C, I) = timed(lambda: pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True))
            info = dict(dbase.LAST_KNN)
            row["flat" if flat else "lists"] = {"sec": t, "tier": info.get("tier"), "flat_terms": info.get("flat_terms"),
                                                "flagged": info.get("flagged"), "path": info.get("path")}
            outs[flat] = (C, I)
        row["equal"] = bool(torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1]))
        res[name] = row
        print(json.dumps({name: row}), flush=True)
        if name == "mixture_s2_k30":
            keep = (X, outs[True][0])
        else:
            del X
        del outs
        torch.cuda.empty_cache()
    # the scan kernel alone, every tile, thresholds = exact k-th distance + the one-term band (what the last pass sees)
    X, C = keep
    L = _lib.lib()
    P = dbase.PackedPoints(X)
    q16, y16, meta = dbase._screen_operands(P, P)
    tau = (C[:, -1] + 2.6).contiguous()
    cap = 256
    buf = torch.empty((n, cap), dtype=torch.int64, device="cuda")
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    n_tiles = (n + 31) // 32
    scans = {}
    shapes = ((1, 0), (2, 0), (3, 0), (1, 10), (3, 10))     # (terms, 0 = natural tile order / 10 = the production pipeline's strided order)
    if len(sys.argv) > 3:
        shapes = tuple((int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[3].split(","))
    stride = 1
    for terms, shape in shapes:
        if shape >= 10:     # shape 1x: the same form with the strided visiting order of the production pipeline
            shape -= 10
            stride = int(n_tiles * 0.6180339887) | 1
            import math
            while math.gcd(stride, n_tiles) != 1:
                stride += 2
        else:
            stride = 1

        def run():
            _lib.check(L.tdr_knn_flat_scan_f32(_lib.ptr(q16), n, 0, _lib.ptr(y16), n, d, terms, 1, 0, n_tiles, stride, _lib.ptr(meta), _lib.ptr(tau),
                                               _lib.ptr(buf), _lib.ptr(cnt), cap, _lib.stream_ptr()), "scan")
        t, _ = timed(run, reps=2)
        flops = 2.0 * n * n * d * terms
        scans[f"terms{terms}_shape{shape}_stride{stride}"] = {"sec": t, "executed_f16_tflops": flops / t / 1e12, "frac_f16_peak": flops / t / 2.5e15,
                                              "mean_appended": float(cnt.float().mean()), "max_appended": int(cnt.max())}
        print(json.dumps({f"scan_terms{terms}_shape{shape}_stride{stride}": scans[f"terms{terms}_shape{shape}_stride{stride}"]}), flush=True)
    res["scan_alone"] = scans
    res["note"] = "one launch over all tiles per (terms, visiting order); the scheduling variants of profiles/r05_knn_flat_variants.json are no longer in the library"
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
