#!/bin/bash
# Round-4 counters of the gradient launch on the bench command (what bounds it: vector issue and the L2's request rate):
# separate rocprofv3 --pmc passes with --kernel-trace only.   gpurun --timeout 900 -- 'bash tools/pmc_grad_r4.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
i=0
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU TCC_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_grad4_$i -- \
      python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-knn-variants --no-configs > $R/gpurun_out/pmc_grad4_$i.log 2>&1
done
cd $R
python - <<'P'
import csv, glob, json
from collections import defaultdict
out = {}
dur = []
for i in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/pmc_grad4_{i}/**/*counter_collection.csv", recursive=True):
        tot, cnt = defaultdict(float), defaultdict(set)
        for r in csv.DictReader(open(f)):
            if "umap_sched_grad_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
        for k in tot:
            out[k] = tot[k] / len(cnt[k])
    for f in glob.glob(f"gpurun_out/pmc_grad4_{i}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "umap_sched_grad_kernel" in r["Kernel_Name"]:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out["avg_launch_ns_under_pmc"] = sum(dur) / max(len(dur), 1)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/r04_grad_pmc.json", "w"), indent=1)
P
