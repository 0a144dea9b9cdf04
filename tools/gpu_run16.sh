#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python $GRAFT_REPO_ROOT/tools/knn_perf.py 1000000 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ -- python $GRAFT_REPO_ROOT/tools/knn_perf.py 1000000 > $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-150
for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | head -1); echo $d; python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if 'knn_scan' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
PY
done
