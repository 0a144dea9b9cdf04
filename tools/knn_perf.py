"""GPU probe: time pack + scan of the exact kNN kernel (HIP events) and print TFLOP/s."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.conftest import gmm
from torchdr_amd.distance import PackedPoints, knn_packed

def run(n, d, k, reps=2):
    X = gmm(n, d, 2.0).cuda()
    torch.cuda.synchronize()
    res = {}
    for r in range(reps + 1):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        P = PackedPoints(X)
        e1.record()
        C, I = knn_packed(P, P, k, "sqeuclidean", True)
        e2.record()
        torch.cuda.synchronize()
        res = {"n": n, "d": d, "k": k, "pack_ms": e0.elapsed_time(e1), "scan_ms": e1.elapsed_time(e2)}
        res["tflops"] = 2.0 * n * n * d / (res["scan_ms"] * 1e-3) / 1e12
    print(json.dumps(res), flush=True)

if __name__ == "__main__":
    # usage: knn_perf.py N [N ...] [d=DIM] [k=K]
    d = next((int(a[2:]) for a in sys.argv[1:] if a.startswith("d=")), 128)
    k = next((int(a[2:]) for a in sys.argv[1:] if a.startswith("k=")), 30)
    for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [100_000]:
        run(n, d, k)
