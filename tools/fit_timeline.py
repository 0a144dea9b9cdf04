"""Timeline of the LAST fit of a traced `tools/fit_time.py` run, between the end of the kNN search and the first gradient launch
(bandwidth search, symmetrisation, renumbering, loop layout, PCA initialisation on its side stream): start offset, duration, queue.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fit_tl -- python $R/tools/fit_time.py 1000000 1
    python tools/fit_timeline.py gpurun_out/fit_tl [min_us]
"""
import csv
import glob
import sys

d = sys.argv[1]
min_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 15e3
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grads = [i for i, r in enumerate(rows) if "umap_pool_grad_kernel" in r["Kernel_Name"]]
# the last fit: the last run of 1000 gradient launches
first_grad = grads[-1000] if len(grads) >= 1000 else grads[0]
t_grad = int(rows[first_grad]["Start_Timestamp"])
win = [r for r in rows[:first_grad + 3] if int(r["Start_Timestamp"]) >= t_grad - 40_000_000]
# start at the previous fit's last gradient launch, if there is one in the window
prev = [i for i, r in enumerate(win) if "umap_pool_grad_kernel" in r["Kernel_Name"] and int(r["Start_Timestamp"]) < t_grad]
if prev:
    win = win[prev[-1] + 1:]
t0 = int(win[0]["Start_Timestamp"])
busy_end = t0
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - busy_end
    if e - s >= min_ns or gap > 30_000:
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} q{r['Queue_Id']:>2} {'(idle %.0f us before) ' % (gap / 1e3) if gap > 30_000 else ''}{r['Kernel_Name'][:90]}")
    busy_end = max(busy_end, e)
