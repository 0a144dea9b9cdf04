#!/bin/bash
# Round-6 closing measurements beside the bench line:   gpurun --timeout 2400 -- 'bash tools/gpu_r6_final.sh'
#   1. one rank's share of a W-rank fit (W = 2, 4, 8; first and middle rank) at N = 1M x 128 and at C4's size (4M x 256)
#      -> gpurun_out/r06_rank_share_{1m,c4}_final.jsonl          (tools/rank_share.py; no link time in these figures)
#   2. the headline fit at other sizes of the generator           -> gpurun_out/r06_sizes_final.jsonl
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -f gpurun_out/r06_rank_share_1m_final.jsonl gpurun_out/r06_rank_share_c4_final.jsonl gpurun_out/r06_sizes_final.jsonl
timeout 700 python tools/rank_share.py --npoints 1000000 --dim 128 --out gpurun_out/r06_rank_share_1m_final.jsonl > /dev/null 2> gpurun_out/r06_rank_share_1m_final.err
python - <<'P'
import json
for l in open("gpurun_out/r06_rank_share_1m_final.jsonl"):
    r = json.loads(l); print(r["n"], r["world"], r["rank"], r["fit_ms"], r.get("speedup_without_link_time"))
P
timeout 1200 python tools/rank_share.py --npoints 4000000 --dim 256 --out gpurun_out/r06_rank_share_c4_final.jsonl > /dev/null 2> gpurun_out/r06_rank_share_c4_final.err
python - <<'P'
import json
for l in open("gpurun_out/r06_rank_share_c4_final.jsonl"):
    r = json.loads(l); print(r["n"], r["world"], r["rank"], r["fit_ms"], r.get("speedup_without_link_time"))
P
for n in 100000 300000 500000 700000 2000000 4000000; do
  timeout 300 python bench.py --npoints $n --steps 3 --warmup 1 --no-cpu-baseline --no-knn-variants --no-configs 2>/dev/null | grep "^{" | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print(json.dumps({'n': $n, 'ms_per_step': round(d['ms_per_step'], 2), 'samples_per_sec': round(d['value']), 'knn_build_ms': round(d['knn_build_sec'] * 1e3, 2), 'phases_ms': {k: round(v, 2) for k, v in d['phases_ms'].items()}, 'knn_path': d.get('knn_path')}))" | tee -a gpurun_out/r06_sizes_final.jsonl | cut -c1-200
done
