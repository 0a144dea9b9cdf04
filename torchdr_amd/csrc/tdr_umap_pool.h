// Launch interface of the pool-sampled UMAP gradient kernel (tdr_umap_pool.hip), shared with the loop object of
// tdr_umap_sched.hip.
#pragma once
#include "tdr_common.h"

// default geometry (geom = 0): threads per block, rows per thread, pool runs, rows per run
#ifndef TDR_POOL_THREADS
#define TDR_POOL_THREADS 512
#endif
#ifndef TDR_POOL_RPT
#define TDR_POOL_RPT 2
#endif
#ifndef TDR_POOL_RUNS
#define TDR_POOL_RUNS 256
#endif
#ifndef TDR_POOL_RUNLEN
#define TDR_POOL_RUNLEN 8
#endif
#define TDR_POOL_NGEOM 6     // tuning geometries 1..6 (TDR_POOL_GEOMS of tdr_umap_pool.hip)
// 0 = default (workgroups per block chosen by the launch size), 1..6 tuning geometries, 16 + s = default with s in {1, 2, 4, 8} workgroups
// per block (s > 1: one row per lane)
static inline bool tdr_pool_geom_ok(int geom) { return (geom >= 0 && geom <= TDR_POOL_NGEOM) || geom == 17 || geom == 18 || geom == 20 || geom == 24; }

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
    float* grad;               // (n_rows, nc), or NULL with Z_out
    float* Z_out;              // non-NULL: the launch also steps its rows, z - lr g -> Z_out (n_total, nc): the other embedding buffer
    float lr;
    const float* lr_table;     // loop object: the learning rate of absolute iteration i is lr_table[i] (then `lr` is unused), and at the
    int check_interval;        //   iterations the reference inspects (i % check_interval == 0) the launch also writes grad, snap (stepped
    float* norm2;              //   rows, (n_rows, nc)) and adds the squared gradient norm to norm2[i / check_interval]
    float* snap;
    int* nan_flag;
    uint32_t n_runs;           // ceil(n_total / rows per run) (set by the launcher)
    int64_t gb0;               // first global row block of the launch (set by the launcher)
    int exact5;                // neg_rate == 5 and n_negatives % 5 == 0: a row's items come in whole groups of five
    unsigned long long* dbg_times;   // measurement only (DBG instances): 8 time stamps per block
    int ablate;                // measurement only: 1 no pool staging, 2 no attraction, 4 no negatives, 8 rows not sorted
};

int launch_pool_grad(const PoolGradParams& P, int geom, hipStream_t st);

}  // namespace tdr
