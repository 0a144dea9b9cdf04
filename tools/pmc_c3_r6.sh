#!/bin/bash
# round 6: rocprofv3 kernel stats + PMC passes (separate --pmc runs with --kernel-trace only) for BASELINE config C3 with the three
# LargeVis samplers of this build -- ne_pull4_runs_kernel (run-permutation, negatives from LDS), ne_pull4_kernel (row permutation,
# every negative gathered), ne_grad_kernel (independent draws + atomics):   gpurun --timeout 900 -- 'bash tools/pmc_c3_r6.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_c3; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/tools/config_roofline.py c3 > $O/roof.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$i -- python $R/tools/config_roofline.py c3 > $O/pmc_$i.log 2>&1
done
cd $R
tail -1 $O/roof.log | cut -c1-400
python - <<PY
import json, subprocess
out = {"source": "tools/pmc_c3_r6.sh on tools/config_roofline.py c3 (LargeVis N = 1M, D = 128, kNN width 15, 5 negatives per row; round-6 build; separate rocprofv3 --pmc passes with --kernel-trace only; per-launch means; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, FETCH_SIZE to be doubled on gfx950 per MI355X_MICROARCH.md)"}
for name, k in (("run-permutation", "ne_pull4_runs_kernel"), ("permutation", "ne_pull4_kernel"), ("independent", "ne_grad_kernel")):
    c = {}
    for i in (1, 2, 3, 4, 5):
        r = subprocess.run(["python", "tools/pmc_sum.py", "$O/pmc_%d" % i, k], capture_output=True, text=True)
        try:
            c.update(json.loads(r.stdout))
        except Exception as e:
            c["error_%d" % i] = (r.stdout + r.stderr)[-300:]
    out[name] = {"kernel": "tdr::" + k, **c}
d = {}
for name in ("run-permutation", "permutation", "independent"):
    c = out[name]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d[name + "_hbm_side_traffic_bytes"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
d["algorithmic_bytes"] = 472000000
out["derived"] = d
json.dump(out, open("$O/r06_c3_pmc.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
f=$(ls -t $O/prof/*/*kernel_stats.csv | head -1); head -8 "$f" | cut -c1-200; cp "$f" $O/r06_c3_kernel_stats.csv
