#!/bin/bash
# Counters of the pool-sampled gradient launch (csrc/tdr_umap_pool.hip) on the N = 1M graph: separate rocprofv3 --pmc passes with
# --kernel-trace only.   gpurun --timeout 900 -- 'bash tools/pmc_pool.sh [geom]'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
G=${1:-0}
mkdir -p $R/gpurun_out
cd /tmp
i=0
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU TCC_BUSY_sum" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PERF_ONLY=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_pool_$i -- \
      python $R/tools/umap_pool_perf.py 1000000 $G > $R/gpurun_out/pmc_pool_$i.log 2>&1
done
rocprofv3 -L 2>/dev/null | grep -o "\b\(TA\|TCP\|TD\|SQ\|TCC\)_[A-Z0-9_a-z]*" | sort -u > $R/gpurun_out/pmc_names.txt
cd $R
python - <<'P'
import csv, glob, json
from collections import defaultdict
out = {}
dur = []
for i in range(1, 11):
    for f in glob.glob(f"gpurun_out/pmc_pool_{i}/**/*counter_collection.csv", recursive=True):
        tot, cnt = defaultdict(float), defaultdict(set)
        for r in csv.DictReader(open(f)):
            if "umap_pool_grad_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
        for k in tot:
            out[k] = tot[k] / len(cnt[k])
    for f in glob.glob(f"gpurun_out/pmc_pool_{i}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "umap_pool_grad_kernel" in r["Kernel_Name"]:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out["avg_launch_ns_under_pmc"] = sum(dur) / max(len(dur), 1)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/r06_pool_pmc.json", "w"), indent=1)
P
