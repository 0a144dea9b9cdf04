#!/bin/bash
# Scratch builds of the library with the scheduled gradient kernel's ablation switches (TDR_GRAD_ABLATE); run HERE.
#   bash tools/grad_ablate.sh "1 2 4 8 15"
cd "$(dirname "$0")/../torchdr_amd/csrc" || exit 1
mkdir -p ../../tools/scratch
for v in ${1:-"1 2 4 8 15"}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DTDR_GRAD_ABLATE=$v -c tdr_umap_sched.hip -o ../../tools/scratch/grad_ab$v.o || exit 1
  objs=$(ls build/*.hip.o | grep -v tdr_umap_sched)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/libtdr_gab$v.so $objs ../../tools/scratch/grad_ab$v.o || exit 1
done
ls -la ../../tools/scratch/libtdr_gab*.so
