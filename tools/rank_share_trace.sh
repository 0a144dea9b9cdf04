#!/bin/bash
# Kernel trace of ONE emulated rank's fit (rank r of W alone on this GPU): which kernels its share consists of.
#   gpurun --timeout 900 -- 'bash tools/rank_share_trace.sh 4000000 256 4 8 c4'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-1000000}; D=${2:-128}; RANK=${3:-4}; W=${4:-8}; TAG=${5:-1m}
mkdir -p $R/gpurun_out
cd /tmp
rm -rf $R/gpurun_out/rs_trace_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rs_trace_$TAG -- \
    python $R/bench.py --emulate-rank $RANK --world $W --npoints $N --dim $D > $R/gpurun_out/rs_trace_$TAG.log 2>&1
cd $R
f=$(find gpurun_out/rs_trace_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r06_rank_share_${TAG}_w${W}_kernel_stats.csv
head -30 gpurun_out/r06_rank_share_${TAG}_w${W}_kernel_stats.csv | cut -c1-200
tail -2 gpurun_out/rs_trace_$TAG.log | cut -c1-1500
