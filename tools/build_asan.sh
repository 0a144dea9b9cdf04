#!/bin/bash
# AddressSanitizer build of the two kernels the optimisation loop and the large searches live in (SURVEY section 5:
# sanitizer coverage): csrc/tdr_umap_sched.hip and csrc/tdr_knn_screen.hip compiled with -fsanitize=address for
# gfx950:xnack+ (device-side ASAN needs page-fault retry), linked with the regular objects of the other files into
# tools/scratch/libtdr_asan.so.  Run HERE (hipcc cross-compiles); tools/asan_smoke.sh runs it on the GPU box.
cd "$(dirname "$0")/../torchdr_amd/csrc" || exit 1
mkdir -p ../../tools/scratch
for f in tdr_umap_sched tdr_knn_screen; do
  hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -g -O2 -std=c++17 -ffp-contract=off -fPIC -c $f.hip -o ../../tools/scratch/$f.asan.o || exit 1
done
objs=$(ls build/*.hip.o | grep -v -e tdr_umap_sched -e tdr_knn_screen)
hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -shared -fPIC -o ../../tools/scratch/libtdr_asan.so $objs ../../tools/scratch/tdr_umap_sched.asan.o ../../tools/scratch/tdr_knn_screen.asan.o || exit 1
ls -la ../../tools/scratch/libtdr_asan.so
