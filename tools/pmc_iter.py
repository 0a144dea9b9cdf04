"""Per-iteration HBM-side traffic of the scheduled UMAP loop from two rocprofv3 --pmc passes over `bench.py` (tools/pmc_bench.sh):
mean FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them) per launch of the gradient, combine + step and schedule-build kernels,
and the per-iteration sum bench.py's `roofline.traffic` is computed from: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes (FETCH_SIZE
doubled on gfx950 per MI355X_MICROARCH.md).

    python tools/pmc_iter.py DIR_FETCH DIR_WRITE > profiles/r04_umap_sched_pmc.json
"""
import csv
import glob
import json
import sys
from collections import defaultdict

# launches per iteration.  Round 6 (pool negatives): the gradient launch carries the SGD step except at the iterations the reference
# inspects (every check_interval-th = 50th), where tdr::sgd_step_kernel runs after it
KERNELS = {"umap_pool_grad_kernel": 1.0, "sgd_step_kernel": 1.0 / 50, "umap_sched_grad_kernel": 1.0, "umap_sched_combine_sgd_kernel": 1.0,
           "umap_sched_build2_kernel": 1.0 / 32, "umap_sched_build_kernel": 1.0 / 32}


def per_launch(d, counter):
    tot, cnt = defaultdict(float), defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for k in KERNELS:
                if k + "<" in r["Kernel_Name"] or k + "(" in r["Kernel_Name"]:
                    tot[k] += float(r["Counter_Value"])
                    cnt[k].add((f, r["Dispatch_Id"]))
    return {k: (tot[k] / len(cnt[k]), len(cnt[k])) for k in tot}


fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
out = {"source": "tools/pmc_bench.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one pass each) over `python bench.py --steps 1 "
                 "--warmup 1 --no-cpu-baseline --no-knn-variants --no-configs` (N = 1M, D = 128, k = 30; both fits counted); KiB per launch as "
                 "rocprofv3 reports them; FETCH_SIZE to be doubled on gfx950 per MI355X_MICROARCH.md",
       "kernels": {}}
F = W = 0.0
for k, share in KERNELS.items():
    if k not in fetch and k not in write:
        continue
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    out["kernels"][k] = {"FETCH_SIZE": f, "WRITE_SIZE": w, "launches_counted": [nf, nw], "launches_per_iteration": share}
    F += f * share
    W += w * share
out["FETCH_SIZE"], out["WRITE_SIZE"] = F, W
out["bytes_per_iteration"] = (2.0 * F + W) * 1024.0
out["FETCH_SIZE_WRITE_SIZE_meaning"] = ("one UMAP iteration of the production loop = the kernels listed under `kernels`, each weighted by its launches per "
                                        "iteration (gradient [+ step] launch, 1/32 of a schedule build); bench.py: roofline.traffic = (2 x FETCH_SIZE + "
                                        "WRITE_SIZE) x 1024")
print(json.dumps(out, indent=1))
