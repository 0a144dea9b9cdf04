"""GPU probe: two-stage (screen + rescore) vs one-stage exact kNN -- time, equality, overflow count."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.conftest import gmm
from torchdr_amd.distance import base as dbase


def run(n, d, k, scale, check=True):
    X = gmm(n, d, scale).cuda()
    out = {"n": n, "d": d, "k": k, "scale": scale}
    res = {}
    for mode in ("force", "0"):
        if mode == "0" and not check:
            continue
        dbase.SCREEN_MODE = mode
        for r in range(2):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            P = dbase.PackedPoints(X)
            if mode == "force":
                P.screen_image()
            e1.record()
            C, I = dbase.knn_packed(P, P, k, "sqeuclidean", True)
            e2.record()
            torch.cuda.synchronize()
        name = "screen" if mode == "force" else "exact"
        out[name + "_pack_ms"] = round(e0.elapsed_time(e1), 3)
        out[name + "_ms"] = round(e1.elapsed_time(e2), 3)
        if mode == "force":
            out["flagged"] = dbase.LAST_KNN["flagged"]
            out["path"] = dbase.LAST_KNN["path"]
            out["tier"] = dbase.LAST_KNN.get("tier")
            out["pruned"] = dbase.LAST_KNN.get("pruned")
        res[name] = (C, I)
    if check:
        out["equal"] = bool(torch.equal(res["screen"][0], res["exact"][0]) and torch.equal(res["screen"][1], res["exact"][1]))
    out["eff_tflops"] = round(2.0 * n * n * d / (out["screen_ms"] * 1e-3) / 1e12, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    # usage: knn_screen_perf.py N [N ...] [d=DIM] [k=K] [s=SCALE] [nocheck]
    d = next((int(a[2:]) for a in sys.argv[1:] if a.startswith("d=")), 128)
    k = next((int(a[2:]) for a in sys.argv[1:] if a.startswith("k=")), 30)
    s = next((float(a[2:]) for a in sys.argv[1:] if a.startswith("s=")), 2.0)
    for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [100_000]:
        run(n, d, k, s, check="nocheck" not in sys.argv)
