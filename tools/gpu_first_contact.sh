#!/bin/bash
# First contact with a node that has more than one MI355X: ONE command that runs everything this build could never run --
#   1. the RCCL tests that every one-GPU box skips (two-rank sharded path, even chunks over 2 / 4 / 8 ranks);
#   2. the peer-write exchange between DISTINCT devices, behind its stress self-check (it falls back to RCCL by itself when HIP IPC
#      or the check fails: the record says which transport ran);
#   3. bench.py --gpus 1 / 2 / 4 / 8 at N = 1M x 128 and at C4's size (4M x 256), with the RCCL all-gather and with the peer exchange;
#   4. the same rank shares measured alone (bench.py --emulate-rank), so that measured / emulated ratios say where the links cost.
# Usage:   bash tools/gpu_first_contact.sh [OUT_DIR]        (one node, no arguments needed; ~10 minutes on 8 GPUs)
# On a box with ONE device every step still runs -- ranks then share the device over gloo / HIP IPC ("devices_shared": true): that is
# how tests/test_distributed_gpu.py::test_first_contact_script_runs_end_to_end exercises this script; the numbers mean nothing there.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/first_contact}
SIZES=${FC_SIZES:-"1000000:128 4000000:256"}
WORLDS=${FC_WORLDS:-"1 2 4 8"}
ITERS=${FC_MAX_ITER:-1000}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
echo "{\"devices\": $NDEV, \"sizes\": \"$SIZES\", \"worlds\": \"$WORLDS\"}" > "$OUT/node.json"
cd "$R"
# 1 + 2: the distributed GPU tests (the RCCL ones run when the node has the devices), the peer exchange forced on
if [ -z "${FC_SKIP_TESTS:-}" ]; then      # (the test that exercises this script sets FC_SKIP_TESTS: it IS one of those tests)
  python -m pytest tests/test_distributed_gpu.py -q -x > "$OUT/tests_distributed.log" 2>&1
  echo "{\"step\": \"tests/test_distributed_gpu.py\", \"rc\": $?}" >> "$OUT/steps.jsonl"
fi
# 3: measured fits
for sz in $SIZES; do
  N=${sz%%:*}; D=${sz##*:}
  for W in $WORLDS; do
    for X in rccl peer; do
      [ "$W" = 1 ] && [ "$X" = peer ] && continue
      FLAG=""; [ "$X" = peer ] && FLAG="--peer-exchange"
      timeout 1200 python bench.py --gpus $W --npoints $N --dim $D --max-iter $ITERS --steps 2 --warmup 1 --no-cpu-baseline --no-knn-variants --no-configs $FLAG \
          > "$OUT/bench_n${N}_w${W}_${X}.json" 2> "$OUT/bench_n${N}_w${W}_${X}.err"
      echo "{\"step\": \"bench n=$N d=$D gpus=$W exchange=$X\", \"rc\": $?}" >> "$OUT/steps.jsonl"
    done
  done
  # 4: the emulated share of a middle rank, for every W > 1
  for W in $WORLDS; do
    [ "$W" = 1 ] && continue
    timeout 1200 python bench.py --emulate-rank $((W / 2)) --world $W --npoints $N --dim $D --max-iter $ITERS > "$OUT/emulated_n${N}_w${W}.json" 2> "$OUT/emulated_n${N}_w${W}.err"
    echo "{\"step\": \"emulated rank share n=$N d=$D world=$W\", \"rc\": $?}" >> "$OUT/steps.jsonl"
  done
done
# summary: one JSON per N
python - "$OUT" <<'P'
import glob, json, os, sys
out = sys.argv[1]
def last_json(path):
    try:
        lines = [l for l in open(path) if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except Exception:
        return None
by_n = {}
for f in sorted(glob.glob(os.path.join(out, "bench_n*_w*_*.json"))):
    rec = last_json(f)
    if not rec:
        continue
    n, w, x = os.path.basename(f)[len("bench_n"):-len(".json")].replace("_w", " ").replace("_", " ").split()
    by_n.setdefault(n, {"measured": [], "emulated": []})["measured"].append({
        "gpus": int(w), "exchange_requested": x, "row_exchange": rec.get("row_exchange"), "devices_shared": rec.get("devices_shared"),
        "ms_per_step": rec.get("ms_per_step"), "samples_per_sec": rec.get("value"), "phases_ms": rec.get("phases_ms"),
        "allgather_us": rec.get("allgather_us"), "knn_build_sec": rec.get("knn_build_sec")})
for f in sorted(glob.glob(os.path.join(out, "emulated_n*_w*.json"))):
    rec = last_json(f)
    if not rec:
        continue
    n, w = os.path.basename(f)[len("emulated_n"):-len(".json")].split("_w")
    e = rec["emulated_rank_share"]
    by_n.setdefault(n, {"measured": [], "emulated": []})["emulated"].append({"world": int(w), "rank": e["rank"], "fit_ms": e["fit_ms"],
        "phases_ms": e["phases_ms"], "exchange_bytes": e.get("exchange_bytes"), "single_process_fit_ms": rec["single_process"]["fit_ms"]})
for n, d in by_n.items():
    one = next((m["ms_per_step"] for m in d["measured"] if m["gpus"] == 1), None)
    for m in d["measured"]:
        m["speedup_vs_one_gpu"] = (one / m["ms_per_step"]) if one and m["ms_per_step"] else None
        e = next((e for e in d["emulated"] if e["world"] == m["gpus"]), None)
        m["measured_over_emulated"] = (m["ms_per_step"] / e["fit_ms"]) if e and m["ms_per_step"] else None
    json.dump(d, open(os.path.join(out, f"first_contact_n{n}.json"), "w"), indent=1)
    print(n, [(m["gpus"], m["exchange_requested"], m.get("row_exchange"), round(m["ms_per_step"] or 0, 1), m["speedup_vs_one_gpu"]) for m in d["measured"]])
P
cat "$OUT/steps.jsonl"
