"""GPU probe: time the kNN scan for the product build and the ablation builds in tools/ablate/."""
import sys, os, json, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.conftest import gmm
from torchdr_amd import _lib

def run(libpath, n, d=128, k=30):
    _lib._lib = None
    _lib.LIB_PATH = libpath
    from torchdr_amd.distance import PackedPoints, knn_packed
    from torchdr_amd.distance import base as dbase
    dbase.SCREEN_MODE = os.environ.get("TDR_KNN_SCREEN", "force")
    X = gmm(n, d, 2.0).cuda()
    P = PackedPoints(X)
    best = 1e9
    for r in range(3):
        e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e1.record(); knn_packed(P, P, k, "sqeuclidean", True); e2.record(); torch.cuda.synchronize()
        best = min(best, e1.elapsed_time(e2))
    print(json.dumps({"lib": os.path.basename(libpath), "n": n, "ms": best, "tflops": 2.0*n*n*d/(best*1e-3)/1e12}), flush=True)

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    base = os.path.join(ROOT, "torchdr_amd", "csrc", "libtorchdr_amd.so")
    for lp in [base] + sorted(glob.glob(os.path.join(ROOT, "tools", "ablate", "*.so"))):
        run(lp, n)
