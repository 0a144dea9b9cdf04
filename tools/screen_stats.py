"""Where the cluster-pruned list-keeping scan spends its wave cycles (measurement build of tdr_knn_screen.hip with -DTDR_SCREEN_STATS).

    python tools/screen_stats.py build          # here (hipcc): torchdr_amd/csrc/build/libtorchdr_amd_stats.so
    python tools/screen_stats.py run            # on the GPU box: headline data, one JSON line per search
"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "torchdr_amd", "csrc")
STATS_LIB = os.path.join(CSRC, "build", "libtorchdr_amd_stats.so")


def build():
    import __graft_entry__ as g

    g.build_hip()
    hipcc = g._hipcc()
    cflags = [f for f in g.HIPCC_FLAGS if f != "-shared"]
    obj = os.path.join(CSRC, "build", "tdr_knn_screen_stats.o")
    subprocess.check_call([hipcc] + cflags + ["-DTDR_SCREEN_STATS", "-c", os.path.join(CSRC, "tdr_knn_screen.hip"), "-o", obj], cwd=CSRC)
    others = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build")))
              if f.endswith(".hip.o") and f != "tdr_knn_screen.hip.o"]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", STATS_LIB, obj] + others, cwd=CSRC)
    print("built", STATS_LIB)


def run(stats=True):
    import time

    import torch

    from torchdr_amd import _lib

    if stats:
        _lib.LIB_PATH = STATS_LIB
    L = _lib.lib()
    if stats:
        L.tdr_debug_screen_stats.restype = ctypes.c_int
        L.tdr_debug_screen_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
    else:
        L.tdr_debug_screen_stats = lambda buf, reset: 0
    from tests.conftest import gmm
    from torchdr_amd.distance import base as B
    from torchdr_amd.distance import pairwise_distances

    n, d, k = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("K", 30))
    X = gmm(n, d, float(os.environ.get("SCALE", 2.0))).cuda()
    buf = (ctypes.c_ulonglong * 8)()
    inner = B._pruned_launch
    scan_ms = [0.0]

    def timed_pruned_launch(*a, **kw):      # counters of the pruned launch alone (the pilots run the same kernel)
        torch.cuda.synchronize()
        L.tdr_debug_screen_stats(buf, 1)
        t0 = time.perf_counter()
        out = inner(*a, **kw)
        torch.cuda.synchronize()
        scan_ms[0] = (time.perf_counter() - t0) * 1e3
        L.tdr_debug_screen_stats(buf, 0)
        return out

    B._pruned_launch = timed_pruned_launch
    for it in range(8):
        lists = 1 - (it // 4) if "LISTS" not in os.environ else int(os.environ["LISTS"])
        L.tdr_knn_screen_clustered_lists(lists)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        s = list(buf)
        print(json.dumps({"lists": "lazy buffers" if lists else "sorted lists", "stats_build": stats, "flagged": B.LAST_KNN.get("flagged"), "search_ms": round(ms, 2), "pruned_scan_and_rescoring_ms": round(scan_ms[0], 2), "path": B.LAST_KNN.get("path"), "tier": B.LAST_KNN.get("tier"),
                          "wave_cycles": s[0], "list_update_cycles": s[1], "list_update_share": round(s[1] / max(s[0], 1), 4),
                          "tile_step_cycles_incl_list_updates": s[7], "tile_step_share": round(s[7] / max(s[0], 1), 4), "barrier_cycles": s[6], "barrier_share": round(s[6] / max(s[0], 1), 4),
                          "merge_events_or_compactions_per_query": round(s[2] / n, 2), "lazy_compaction_cycles_each": round(s[3] / max(s[2], 1), 1) if lists else None, "lazy_bisection_steps_each": round(s[4] / max(s[2], 1), 2) if lists else None,
                          "merged_survivors_per_query": None if lists else round(s[3] / n, 2), "serial_insertions_per_query": None if lists else round(s[4] / n, 2), "tile_steps_per_wave_tile": round(s[5] / (n / 32), 2),
                          "note": "counters of the pruned launch alone; the timers add a few per cent to the scan"}), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run(stats=sys.argv[1] != "time")
