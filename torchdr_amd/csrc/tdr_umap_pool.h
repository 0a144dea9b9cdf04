// Launch interface of the pool-sampled UMAP gradient kernel (tdr_umap_pool.hip), shared with the loop object of
// tdr_umap_sched.hip.
#pragma once
#include "tdr_common.h"

// default geometry (geom = 0): rows per workgroup / pool runs of 16 rows
#ifndef TDR_POOL_ROWS
#define TDR_POOL_ROWS 512
#endif
#ifndef TDR_POOL_RUNS
#define TDR_POOL_RUNS 256
#endif

namespace tdr {

struct PoolGradParams {
    const float* Z;
    int nc;
    int64_t n_total, row0, n_rows;
    const int32_t* list;
    const uint2* hdr;          // (B, n_rows) records of a ONE-slice schedule
    int t_local;
    float a, b;
    int neg_rate, n_negatives;
    uint64_t seed;
    uint32_t iter;
    const int* iter_base;      // optional device int added to iter (graph replays)
    float exag, rep, eps;
    float* grad;               // (n_rows, nc)
    uint32_t n_runs;           // ceil(n_total / 16)
    int64_t gb0;               // first global row block of the launch (set by the launcher)
};

int launch_pool_grad(const PoolGradParams& P, int geom, hipStream_t st);

}  // namespace tdr
