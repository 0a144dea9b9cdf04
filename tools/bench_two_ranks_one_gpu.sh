#!/bin/bash
# bench.py's N > 1 control flow on a ONE-GPU box: two ranks share device 0, gloo backend (host-staged collectives).
# Checks the sharded path end to end (pruned sharded kNN, row exchange, per-iteration all-gather, max-over-ranks timing,
# rank-0 JSON line); the timing itself is meaningless.   gpurun -- 'bash tools/bench_two_ranks_one_gpu.sh [N]'
N=${1:-200000}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 WORLD_SIZE=2 LOCAL_RANK=0 TDR_DIST_BACKEND=gloo
mkdir -p gpurun_out
RANK=1 timeout 280 python bench.py --gpus 2 --npoints $N --steps 1 --warmup 1 --max-iter 200 --no-cpu-baseline > gpurun_out/bench_2r_rank1.log 2>&1 &
P1=$!
RANK=0 timeout 280 python bench.py --gpus 2 --npoints $N --steps 1 --warmup 1 --max-iter 200 --no-cpu-baseline > gpurun_out/bench_2r_rank0.log 2>&1
wait $P1
tail -2 gpurun_out/bench_2r_rank1.log | cut -c1-300
tail -1 gpurun_out/bench_2r_rank0.log | cut -c1-900
