#!/bin/bash
# gpurun --timeout 400 -- 'bash tools/asan_smoke.sh'    (after `bash tools/build_asan.sh` in the build container)
# Smoke-size runs of the scheduled UMAP loop and of the two-stage / pruned kNN search on the ASAN build; the report (or
# "no errors") goes to gpurun_out/asan_smoke.log.  Bounded by its own timeout: device ASAN needs XNACK, which a box may refuse.
export HSA_XNACK=1
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
mkdir -p gpurun_out
LD_PRELOAD=$RT timeout 240 python tools/asan_smoke.py > gpurun_out/asan_smoke.log 2>&1
echo "exit $?" >> gpurun_out/asan_smoke.log
tail -15 gpurun_out/asan_smoke.log
