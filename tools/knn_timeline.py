"""Timeline of one kNN-graph build at the headline size (N = 1M, D = 128, k = 30): which kernels are on the critical path and
where the device idles (host reads of pilot results).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/knn_tl -- python $R/tools/knn_timeline.py run
    python tools/knn_timeline.py parse gpurun_out/knn_tl        # prints the LAST search's kernels in start order
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import time

    import torch

    from tests.conftest import gmm
    from torchdr_amd.distance import pairwise_distances
    n, d, k = int(os.environ.get("N", 1_000_000)), 128, int(os.environ.get("K", 30))
    X = gmm(n, d, 2.0).cuda()
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marker = torch.zeros(1, device="cuda").fill_(float(it))      # a fill kernel marks the start of each search
        pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True)
        torch.cuda.synchronize()
        print("search", it, "ms", (time.perf_counter() - t0) * 1e3, flush=True)


def parse(d, span_ms=33.5):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last search = the kernels inside the last `span_ms` before the final kernel's end
    last_end = max(int(r["End_Timestamp"]) for r in rows)
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= last_end - int(span_ms * 1e6)]
    t0 = int(rows[0]["Start_Timestamp"])
    end_prev = t0
    busy_until = t0
    idle = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = s - busy_until
        if gap > 0:
            idle += gap
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
        print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  gap {max(gap, 0) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
        busy_until = max(busy_until, e)
    print(f"span {(busy_until - t0) / 1e6:.3f} ms, device idle inside it {idle / 1e6:.3f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 33.5)
