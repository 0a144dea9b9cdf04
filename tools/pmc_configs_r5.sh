#!/bin/bash
# round 5: rocprofv3 kernel stats + PMC passes for BASELINE configs C3 (ne_grad_kernel) and C5 (pair_scan_kernel<SeaStats>) with the
# CURRENT kernels (VERDICT r04 #2b):  gpurun --timeout 900 -- 'bash tools/pmc_configs_r5.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_cfg; mkdir -p $O
cd /tmp
for c in c3 c5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python $R/tools/config_roofline.py $c > $O/roof_$c.log 2>&1
done
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  for c in c3 c5; do
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_${c}_$i -- python $R/tools/config_roofline.py $c > $O/pmc_${c}_$i.log 2>&1
  done
done
cd $R
for c in c3 c5; do tail -1 $O/roof_$c.log | cut -c1-600; done
for c in c3 c5; do
  k=ne_grad_kernel; [ $c = c5 ] && k=pair_scan_kernel
  python - <<PY
import json, subprocess, glob
out = {}
for i in (1, 2, 3, 4):
    r = subprocess.run(["python", "tools/pmc_sum.py", "$O/pmc_${c}_%d" % i, "$k"], capture_output=True, text=True)
    out.update(json.loads(r.stdout))
json.dump(out, open("$O/${c}_pmc_raw.json", "w"), indent=1)
print("$c", json.dumps(out)[:900])
PY
  f=$(ls -t $O/prof_$c/*/*kernel_stats.csv | head -1); head -6 "$f" | cut -c1-200; cp "$f" $O/${c}_kernel_stats.csv
done
