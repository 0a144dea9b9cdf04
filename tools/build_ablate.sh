#!/bin/bash
# Scratch builds of the library with the schedule kernel's ablation switches (tools/sched_build_ablate.py); run HERE
# (hipcc cross-compiles), the .so files travel to the GPU box with the snapshot (git-ignored).
#   bash tools/build_ablate.sh "1 3 4 8 16"
cd "$(dirname "$0")/../torchdr_amd/csrc" || exit 1
mkdir -p ../../tools/scratch
for v in ${1:-"1 3 4 8 16"}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DTDR_SCHED_ABLATE=$v -c tdr_umap_sched.hip -o ../../tools/scratch/sched_ab$v.o || exit 1
  objs=$(ls build/*.hip.o | grep -v tdr_umap_sched)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/libtdr_ab$v.so $objs ../../tools/scratch/sched_ab$v.o || exit 1
done
ls -la ../../tools/scratch/*.so
