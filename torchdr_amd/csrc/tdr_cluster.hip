// Coarse cluster index of the pruned exact self search (DESIGN.md "Cluster-bound pruning"), built with the package's own
// kernels.  The search result never depends on this index -- only the number of tiles the scan may skip does -- so the
// index is free to be approximate (sampling, projected seeding); its BOUNDS are not: radii are
// rounded up and centre distances down from direct-difference evaluations.
//
// Stages (host: torchdr_amd/distance/base.py:ClusterIndex):
//   1. sample      S = 16 C (<= 16384) stratified rows, gathered in full dimension
//   2. seeding     farthest-point (max-min) selection of C seeds on the EXACT squared distances of the sample: the S x S
//                  matrix comes from the dense MFMA kernel (tdr_dense_dist_packed_f32, 17-70 GFLOP), then ONE workgroup
//                  walks it -- a step reads the newest seed's row (<= 64 KiB, contiguous), updates the running
//                  min-distances it keeps in registers and takes a workgroup arg-max: no launches, no grid barrier (the
//                  previous version launched one kernel per seed: 1000 launches, 10.4 ms of an N = 1M fit; a 16-d
//                  random projection small enough for registers was tried first and mis-seeds ~1 % of the blobs, which
//                  costs the pruned scan 5x)
//   3. Lloyd       nearest centre of the sample / of all N points through the exact kNN kernel with k = 1 (K1), centroid
//                  sums in sample order (one workgroup per cluster)
//   4. bounds      radius_c = max |x - c| over the members (direct difference, rounded up), centre distance matrix
//                  (direct difference, rounded down), visiting order = per-row rank sort of the centre distances
//   5. layout      members of a cluster contiguous and in ascending row order (stable counting sort: deterministic), every
//                  cluster padded to a multiple of 32 rows (row_map, -1 = padding); + the compact order perm / inv
#include "tdr_common.h"

namespace tdr {

__device__ __forceinline__ uint32_t cmix32(uint32_t x) {
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
    return x;
}

constexpr int CL_PP = 16;    // sample points per thread of the seeding workgroup
constexpr int CL_TH = 1024;  // threads of the seeding workgroup

// ---- 1. stratified sample: stratum s covers rows [s n / S, (s + 1) n / S); one hashed row of it -----------------------
__global__ __launch_bounds__(256) void cluster_sample_kernel(int64_t n, int S, uint32_t seed, int32_t* __restrict__ sample_idx) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int64_t lo = (int64_t)s * n / S, hi = (int64_t)(s + 1) * n / S;
    sample_idx[s] = (int32_t)(lo + (hi > lo ? (int64_t)(cmix32(seed ^ (uint32_t)s * 0x9E3779B9u) % (uint32_t)(hi - lo)) : 0));
}

// ---- 2. max-min seeding on the sample's distance matrix, one workgroup ------------------------------------------------
// Adaptive form (c_min < C, n_out != NULL): the max-min distance delta_t of the t-th seed is non-increasing; on data made of
// well-separated groups it collapses once every group holds a seed (the next farthest point lies INSIDE a group).  The loop
// stops at the first step t >= c_min whose delta falls below `drop` x the previous one and reports t seeds: one ball per
// group, whatever the number of groups between c_min and C.  No such step: c_min seeds (the fixed default).
__global__ __launch_bounds__(CL_TH) void cluster_maxmin_kernel(const float* __restrict__ D2, int64_t ld, int S, int C,
                                                              int32_t* __restrict__ seeds, int c_min, float drop,
                                                              int32_t* __restrict__ n_out) {
    __shared__ unsigned long long wbest[CL_TH / 64];
    __shared__ int winner;
    __shared__ int stop_at;
    if (threadIdx.x == 0) stop_at = -1;
    float prev_delta = 3.0e38f;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float mind[CL_PP];
#pragma unroll
    for (int p = 0; p < CL_PP; ++p) mind[p] = (p * CL_TH + tid) < S ? 3.0e38f : -1.0f;
    int cur = 0;  // first seed: sample point 0
    for (int step = 0; step < C; ++step) {
        if (tid == 0) seeds[step] = cur;
        const float* row = D2 + (size_t)cur * ld;
        float best = -2.0f;
        int bi = 0;
#pragma unroll
        for (int p = 0; p < CL_PP; ++p) {
            const int i = p * CL_TH + tid;
            if (i < S) mind[p] = fminf(mind[p], row[i]);
            if (mind[p] > best) { best = mind[p]; bi = i; }
        }
        // workgroup arg-max of (min-distance, index): clamped at 0, so the bit patterns order like the values
        unsigned long long key = ((unsigned long long)__float_as_uint(fmaxf(best, 0.f)) << 32) | (unsigned)(0x7fffffff - bi);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o, 64);
            key = other > key ? other : key;
        }
        if (lane == 0) wbest[w] = key;
        __syncthreads();
        if (w == 0) {       // the 16 per-wavefront maxima: one more butterfly in the first wavefront (was a serial scan by thread 0)
            unsigned long long k = lane < CL_TH / 64 ? wbest[lane] : 0ull;
#pragma unroll
            for (int o = CL_TH / 128; o > 0; o >>= 1) {
                const unsigned long long other = __shfl_xor(k, o, 64);
                k = other > k ? other : k;
            }
            if (lane == 0) {
                winner = 0x7fffffff - (int)(unsigned)(k & 0xffffffffu);
                const float delta = __uint_as_float((unsigned)(k >> 32));     // max-min distance of the NEXT seed
                if (n_out && step + 1 >= c_min && step > 0 && delta < drop * prev_delta) stop_at = step + 1;
                prev_delta = delta;
            }
        }
        __syncthreads();
        cur = winner;
        if (stop_at >= 0) break;
    }
    if (tid == 0 && n_out) *n_out = stop_at >= 0 ? stop_at : c_min;
}

// ---- 2b. the same seeding, TWO seeds per dependent step (round 6) -----------------------------------------------------------
// A step of cluster_maxmin_kernel is one dependent chain -- arg-max -> read that row of D2 (32 KiB of a 256 MB matrix: 2-3 us of
// latency) -> update -> arg-max: 3.2 us per seed alone, 4.5 us next to the pilots of a kNN build (4.5 ms at 1000 seeds, the longest
// link of the build's critical path).  Adding the winner w lowers the min-distances only around it: if the RUNNER-UP r of the same
// arg-max is at least as far from w as from the seeds so far (D2[w][r] >= mind[r] > 0), then after the update mind[r] is unchanged
// and every other point's value has not grown -- r is exactly the next greedy pick (keys are distinct: (value, smaller index
// first)).  So a step finds winner AND runner-up, reads both rows (and D2[w][r]) at once, and takes the runner-up as the following
// seed when that test holds; when it fails, only the winner is taken.  The seeds are those of the one-chain kernel bit for bit,
// fixed and adaptive count.  (Round 6 first tried PREFETCHING the runner-up's row under the next arg-max: one arg-max per seed
// stayed on the chain and the extra registers cost more than the overlap gained -- 3.30 vs 3.17 ms, dropped.)
// Reductions by DPP row shifts (no LDS round trips); one barrier per arg-max, the per-wavefront slots double-buffered.
__device__ __forceinline__ unsigned long long dpp_max_step_u64(unsigned long long v, unsigned long long o) { return o > v ? o : v; }
#define TDR_DPP_U64(V, CTRL, RM)                                                                                         \
    {                                                                                                                    \
        const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(V), (int)(unsigned)(V), CTRL, RM, 0xf, false);                 \
        const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)((V) >> 32), (int)(unsigned)((V) >> 32), CTRL, RM, 0xf, false); \
        V = dpp_max_step_u64(V, ((unsigned long long)hi_ << 32) | lo_);                                                  \
    }
// max over the wavefront, valid in lane 63
__device__ __forceinline__ unsigned long long wave_max_u64_lane63(unsigned long long v) {
    TDR_DPP_U64(v, 0x111, 0xf) TDR_DPP_U64(v, 0x112, 0xf) TDR_DPP_U64(v, 0x114, 0xf) TDR_DPP_U64(v, 0x118, 0xf)
    TDR_DPP_U64(v, 0x142, 0xa) TDR_DPP_U64(v, 0x143, 0xc)
    return v;
}
// max over the 16 lanes of a row (lanes 0..15 hold the per-wavefront values), valid in lane 15
__device__ __forceinline__ unsigned long long row_max_u64_lane15(unsigned long long v) {
    TDR_DPP_U64(v, 0x111, 0xf) TDR_DPP_U64(v, 0x112, 0xf) TDR_DPP_U64(v, 0x114, 0xf) TDR_DPP_U64(v, 0x118, 0xf)
    return v;
}
__device__ __forceinline__ unsigned long long readlane_u64c(unsigned long long v, int l) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
}

__global__ __launch_bounds__(CL_TH) void cluster_maxmin2_kernel(const float* __restrict__ D2, int64_t ld, int S, int C,
                                                               int32_t* __restrict__ seeds, int c_min, float drop,
                                                               int32_t* __restrict__ n_out) {
    constexpr int NWV = CL_TH / 64;
    static_assert(NWV == 16, "one DPP row of per-wavefront maxima");
    __shared__ unsigned long long wslot[4][NWV];     // [parity of the step][winner / runner-up pass]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float mind[CL_PP];
#pragma unroll
    for (int p = 0; p < CL_PP; ++p) mind[p] = 3.0e38f;
    // workgroup arg-max of (min-distance clamped at 0, smaller index first), the entry `skip` left out; the same value in every thread
    auto argmax = [&](int slot, int skip) -> unsigned long long {
        unsigned long long key = 0ull;     // below every real key (index part >= 0x7fffffff - S > 0)
#pragma unroll
        for (int p = 0; p < CL_PP; ++p) {
            const int i = p * CL_TH + tid;
            if (i < S && i != skip) {
                const unsigned long long k = ((unsigned long long)__float_as_uint(fmaxf(mind[p], 0.f)) << 32) | (unsigned)(0x7fffffff - i);
                key = k > key ? k : key;
            }
        }
        key = wave_max_u64_lane63(key);
        if (lane == 63) wslot[slot][w] = key;
        __syncthreads();
        unsigned long long g = lane < NWV ? wslot[slot][lane] : 0ull;
        g = row_max_u64_lane15(g);
        return readlane_u64c(g, 15);
    };
    int n_seeds = 1, stop_at = -1;
    float prev_delta = 3.0e38f;
    if (tid == 0) seeds[0] = 0;
    {   // the first seed: sample point 0
        const float* row = D2;
#pragma unroll
        for (int p = 0; p < CL_PP; ++p) { const int i = p * CL_TH + tid; if (i < S) mind[p] = fminf(mind[p], row[i]); }
    }
    int par = 0;
    while (n_seeds < C) {
        // winner: the next seed (its key's value part = the max-min distance at which it is taken)
        const unsigned long long kw = argmax(2 * par, -1);
        const int wi = 0x7fffffff - (int)(unsigned)(kw & 0xffffffffu);
        const float dw = __uint_as_float((unsigned)(kw >> 32));
        if (n_out && n_seeds >= c_min && n_seeds > 1 && dw < drop * prev_delta) { stop_at = n_seeds; break; }
        prev_delta = dw;
        const float* rowW = D2 + (size_t)wi * ld;
        float rw[CL_PP];
#pragma unroll
        for (int p = 0; p < CL_PP; ++p) { const int i = p * CL_TH + tid; rw[p] = i < S ? rowW[i] : 0.f; }
        if (tid == 0) seeds[n_seeds] = wi;
        ++n_seeds;
        if (n_seeds >= C) {
#pragma unroll
            for (int p = 0; p < CL_PP; ++p) mind[p] = fminf(mind[p], rw[p]);
            break;
        }
        // runner-up of the same arg-max (the winner's row is on its way)
        const unsigned long long kr = argmax(2 * par + 1, wi);
        par ^= 1;
        const int ri = 0x7fffffff - (int)(unsigned)(kr & 0xffffffffu);
        const float dr = __uint_as_float((unsigned)(kr >> 32));
        const bool have_runner = kr != 0ull && dr > 0.f;
        float rr[CL_PP];
        float dwr = -1.f;
        if (have_runner) {
            const float* rowR = D2 + (size_t)ri * ld;
            dwr = rowW[ri];
#pragma unroll
            for (int p = 0; p < CL_PP; ++p) { const int i = p * CL_TH + tid; rr[p] = i < S ? rowR[i] : 0.f; }
        }
#pragma unroll
        for (int p = 0; p < CL_PP; ++p) mind[p] = fminf(mind[p], rw[p]);
        if (have_runner && dwr >= dr) {
            // the runner-up is the next greedy pick: its own stopping test, then its row
            if (n_out && n_seeds >= c_min && dr < drop * prev_delta) { stop_at = n_seeds; break; }
            prev_delta = dr;
#pragma unroll
            for (int p = 0; p < CL_PP; ++p) mind[p] = fminf(mind[p], rr[p]);
            if (tid == 0) seeds[n_seeds] = ri;
            ++n_seeds;
        }
    }
    if (n_out && stop_at < 0 && n_seeds >= C && C >= c_min && C > 1) {
        // the one-chain kernel also tests the distance that a (C + 1)-th seed would have
        const unsigned long long kw = argmax(2 * par, -1);
        if (__uint_as_float((unsigned)(kw >> 32)) < drop * prev_delta) stop_at = C;
    }
    if (tid == 0 && n_out) *n_out = stop_at >= 0 ? stop_at : c_min;
}
#undef TDR_DPP_U64

// ---- predicted scan share of a pruned search (tdr_cluster_scan_fraction_f32): one workgroup per query cluster -------------
__global__ __launch_bounds__(256) void scan_fraction_kernel(const float* __restrict__ dist, const float* __restrict__ radius,
                                                            const int32_t* __restrict__ tiles, int C, float tau,
                                                            unsigned long long* __restrict__ out) {
    __shared__ unsigned long long red[4];
    const int w = blockIdx.x;
    const float rw = radius[w];
    unsigned long long s = 0ull;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float gap = fmaxf(dist[(size_t)w * C + c] - rw - radius[c], 0.f);
        if (gap * gap <= tau) s += (unsigned long long)tiles[c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long v = red[0] + red[1] + red[2] + red[3];
        const unsigned long long tw = (unsigned long long)tiles[w];
        atomicAdd(&out[0], v * tw);
        atomicAdd(&out[1], tw);     // out[1] = sum of tiles (squared by the caller)
    }
}

// ---- per-tile lower bound of the distance to every centre (tdr_cluster_tile_cdist_f32) -----------------------------------
__global__ __launch_bounds__(256) void tile_cdist_kernel(const float* __restrict__ d2, int64_t ld, int C, float eps,
                                                         const int32_t* __restrict__ row_map, const float* __restrict__ xn,
                                                         const float* __restrict__ cn, float* __restrict__ out) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.x * 32;
    const float cnc = cn[c];
    float best = __builtin_inff();
    for (int i = 0; i < 32; ++i) {
        if (row_map[r0 + i] < 0) continue;
        const float v = d2[(size_t)(r0 + i) * ld + c] - eps * (xn[r0 + i] + cnc);
        best = fminf(best, v);
    }
    float t = best;
    if (best < __builtin_inff()) {
        t = sqrtf(fmaxf(best, 0.f));     // round to nearest; the factor below takes it down by more than an ulp
        t = t * 0.999999f;
    }
    out[(size_t)blockIdx.x * C + c] = t;
}

// ---- gather rows: out[i] = X[idx[i]] (optionally through a second index: X[idx[idx2[i]]]) ---------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ X, int64_t ldx, int d,
                                                          const int32_t* __restrict__ idx, const int32_t* __restrict__ idx2,
                                                          int64_t m, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m * d) return;
    const int64_t i = e / d;
    const int c = (int)(e - i * d);
    const int64_t row = idx[idx2 ? idx2[i] : i];
    out[e] = X[row * ldx + c];
}

// ---- 3. Lloyd update: cent[l] = mean of the sample points labelled l.  One workgroup per cluster, one thread per feature
// column, members added in sample order: the centres -- and with them the assignments and the cluster-sorted order that
// callers renumber their points by -- are the same on every run (atomics would make the last bits depend on arrival order).
// The workgroup tests 256 labels at a time (one ballot per wavefront, double-buffered in LDS: one barrier per chunk) and
// walks the set bits in order; a cluster has ~8 members among the 8 C samples, so most chunks cost the test alone.
__global__ __launch_bounds__(256) void centroid_update_kernel(const float* __restrict__ Xs, int64_t S, int d,
                                                              const int32_t* __restrict__ labels, float* __restrict__ cent) {
    __shared__ unsigned long long wmask[2][4];
    const int l = blockIdx.x, w = threadIdx.x >> 6;
    for (int c0 = 0; c0 < d; c0 += 1024) {
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
        int n = 0, it = 0;
        for (int64_t s0 = 0; s0 < S; s0 += 256, ++it) {
            const int64_t s = s0 + threadIdx.x;
            const bool hit = s < S && labels[s] == l;
            const unsigned long long m = __ballot(hit);
            if ((threadIdx.x & 63) == 0) wmask[it & 1][w] = m;
            __syncthreads();
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                unsigned long long mm = wmask[it & 1][ww];
                while (mm) {
                    const int bpos = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    const float* row = Xs + (size_t)(s0 + ww * 64 + bpos) * d;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = c0 + q * 256 + threadIdx.x;
                        if (c < d) sum[q] += row[c];
                    }
                    ++n;
                }
            }
        }
        __syncthreads();  // the next column group starts over with buffer 0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + q * 256 + threadIdx.x;
            if (c < d && n > 0) cent[(size_t)l * d + c] = sum[q] / (float)n;  // an empty cluster keeps its centre
        }
    }
}

// ---- 4. bounds ----------------------------------------------------------------------------------------------------------
// radius[label] = max |x - c_label| (direct difference) and the cluster sizes; one wavefront per point
__global__ __launch_bounds__(256) void cluster_radius_kernel(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                             const int32_t* __restrict__ labels, const float* __restrict__ cent,
                                                             unsigned* __restrict__ radius_bits, int32_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int l = labels[r];
    float s = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float t = X[r * ldx + c] - cent[(size_t)l * d + c];
        s += t * t;
    }
    s = wave_sum(s);
    if (lane == 0) {
        atomicMax(&radius_bits[l], __float_as_uint(sqrtf(s)));  // >= 0: bit order = value order
        atomicAdd(&counts[l], 1);
    }
}
// radius *= 1 + 1e-5 (+ tiny): the bound has to dominate the rounding of the sums above
__global__ void radius_round_up_kernel(float* __restrict__ radius, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) radius[c] = radius[c] * (1.0f + 1e-5f) + 1e-30f;
}

// centre distance matrix (direct difference, rounded down) and, per row, the clusters by increasing centre distance
// (rank sort in LDS; C <= 4096)
__global__ __launch_bounds__(256) void centre_tables_kernel(const float* __restrict__ cent, int C, int d, float* __restrict__ dist,
                                                            int32_t* __restrict__ order) {
    extern __shared__ float row[];  // C distances
    const int a = blockIdx.x;
    for (int b = threadIdx.x; b < C; b += 256) {
        float s = 0.f;
        for (int c = 0; c < d; ++c) {
            const float t = cent[(size_t)a * d + c] - cent[(size_t)b * d + c];
            s += t * t;
        }
        row[b] = sqrtf(s);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < C; b += 256) {
        const float mine = row[b];
        int rank = 0;
        for (int q = 0; q < C; ++q) {
            const float o = row[q];
            rank += (o < mine || (o == mine && q < b)) ? 1 : 0;
        }
        order[(size_t)a * C + rank] = b;
        dist[(size_t)a * C + b] = mine * (1.0f - 1e-5f);
    }
}

// ---- 5. layout ------------------------------------------------------------------------------------------------------------
// one workgroup: tiles_c = ceil(count_c / 32), tile_begin = exclusive scan, tile_cluster[t] = c, *n_img = 32 * total tiles
__global__ __launch_bounds__(256) void cluster_tiles_kernel(const int32_t* __restrict__ counts, int C, int32_t* __restrict__ tile_begin,
                                                            int32_t* __restrict__ tiles, int32_t* __restrict__ tile_cluster,
                                                            int64_t* __restrict__ n_img, int32_t* __restrict__ row_begin) {
    __shared__ int tot[256];
    __shared__ int rtot[256];
    const int tid = threadIdx.x;
    const int per = (C + 255) / 256;
    const int c0 = tid * per, c1 = (c0 + per < C) ? c0 + per : C;
    int s = 0, rs = 0;
    for (int c = c0; c < c1; ++c) { s += (counts[c] + 31) / 32; rs += counts[c]; }
    tot[tid] = s;
    rtot[tid] = rs;
    __syncthreads();
    if (tid == 0) {
        int run = 0, rrun = 0;
        for (int i = 0; i < 256; ++i) {
            const int t = tot[i]; tot[i] = run; run += t;
            const int rt = rtot[i]; rtot[i] = rrun; rrun += rt;
        }
        tile_begin[C] = run;
        row_begin[C] = rrun;
        *n_img = (int64_t)run * 32;
    }
    __syncthreads();
    int run = tot[tid], rrun = rtot[tid];
    for (int c = c0; c < c1; ++c) {
        const int t = (counts[c] + 31) / 32;
        tile_begin[c] = run;
        row_begin[c] = rrun;
        tiles[c] = t;
        for (int i = 0; i < t; ++i) tile_cluster[run + i] = c;
        run += t;
        rrun += counts[c];
    }
}
// Stable layout: the members of a cluster in ASCENDING ROW ORDER -- the same on every run and on every rank of a row-sharded
// fit (each rank builds the index itself instead of receiving it; an atomic cursor per cluster would hand out positions in
// arrival order).  A counting sort by label in three steps: the rows are cut into NB <= 2048 runs of `rps` consecutive rows,
// one wavefront per run; (1) H[c][b] = members of cluster c in run b, (2) exclusive scan of every H[c][.], (3) a run walks
// its rows 64 at a time: lanes holding the same label rank themselves by lane (ballots), the group's first lane advances
// H[c][b] with ONE returning atomic and the others take its value -- a run is touched by one wavefront only, in order.
// Also emits the compact cluster-sorted order (perm: position -> row, inv: row -> position, ppos: position -> padded position).
__global__ __launch_bounds__(256) void cluster_hist_kernel(const int32_t* __restrict__ labels, int64_t n, int rps, int NB,
                                                           int32_t* __restrict__ H) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= NB) return;
    const int64_t r1 = ((int64_t)(b + 1) * rps < n) ? (int64_t)(b + 1) * rps : n;
    for (int64_t r = (int64_t)b * rps + lane; r < r1; r += 64) atomicAdd(&H[(size_t)labels[r] * NB + b], 1);
}
// one wavefront per cluster: H[c][.] <- exclusive prefix sums over the runs
__global__ __launch_bounds__(256) void cluster_hist_scan_kernel(int32_t* __restrict__ H, int C, int NB) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    int32_t* h = H + (size_t)c * NB;
    const int per = (NB + 63) / 64;
    const int b0 = lane * per, b1 = (b0 + per < NB) ? b0 + per : NB;
    int s = 0;
    for (int b = b0; b < b1; ++b) s += h[b];
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    int run = incl - s;
    for (int b = b0; b < b1; ++b) { const int t = h[b]; h[b] = run; run += t; }
}
__global__ __launch_bounds__(256) void cluster_scatter_kernel(const int32_t* __restrict__ labels, int64_t n, int rps, int NB,
                                                              const int32_t* __restrict__ tile_begin, const int32_t* __restrict__ row_begin,
                                                              int32_t* __restrict__ H, int32_t* __restrict__ row_map,
                                                              int32_t* __restrict__ perm, int32_t* __restrict__ inv,
                                                              int32_t* __restrict__ ppos) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= NB) return;
    const int64_t r1 = ((int64_t)(b + 1) * rps < n) ? (int64_t)(b + 1) * rps : n;
    for (int64_t r0 = (int64_t)b * rps; r0 < r1; r0 += 64) {
        const int64_t r = r0 + lane;
        const bool valid = r < r1;
        const int l = valid ? labels[r] : -1 - lane;   // idle lanes: labels of their own
        int rank = 0, size = 1, leader = lane;
        unsigned long long rem = __ballot(true);
        while (rem) {
            const int ld = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rem));
            const int lbl = __builtin_amdgcn_readlane(l, ld);
            const unsigned long long m = __ballot(l == lbl);
            if (l == lbl) {
                rank = __popcll(m & ((1ull << lane) - 1ull));
                size = __popcll(m);
                leader = ld;
            }
            rem &= ~m;
        }
        int old = 0;
        if (valid && lane == leader) old = atomicAdd(&H[(size_t)l * NB + b], size);
        old = __shfl(old, leader, 64);
        if (valid) {
            const int pos = old + rank;
            const int pp = tile_begin[l] * 32 + pos;
            row_map[pp] = (int32_t)r;
            if (perm) {
                const int cp = row_begin[l] + pos;
                perm[cp] = (int32_t)r;
                inv[r] = cp;
                ppos[cp] = pp;
            }
        }
    }
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* 1. indices of S stratified sample rows of an n-row block. */
int tdr_cluster_sample_i32(int64_t n, int S, uint32_t seed, int32_t* sample_idx, void* stream) {
    if (!sample_idx || n <= 0 || S <= 0 || n >= 0x7fffffffLL) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(cluster_sample_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, S, seed, sample_idx);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* largest sample the seeding workgroup handles */
static int g_maxmin_two = 1;   // 0: the one-chain kernel of rounds 2-5 (tdr_cluster_maxmin_mode: measurements, equality test)
int tdr_cluster_maxmin_capacity(void) { return CL_PP * CL_TH; }
/* Measurement / test switch of the seeding: 1 (default) = winner and runner-up per dependent step (the runner-up becomes the next
 * seed when it is at least as far from the winner as from the seeds before), 0 = one seed per step; the same seeds either way.
 * Returns the previous value. */
int tdr_cluster_maxmin_mode(int two_per_step) {
    const int old = g_maxmin_two;
    if (two_per_step == 0 || two_per_step == 1) g_maxmin_two = two_per_step;
    return old;
}

/* 2. C farthest-point seeds (indices into the sample) from the sample's S x S squared-distance matrix D2 (row stride ld);
 * S <= tdr_cluster_maxmin_capacity(). */
int tdr_cluster_maxmin_f32(const float* D2, int64_t ld, int S, int C, int32_t* seeds, void* stream) {
    if (!D2 || !seeds || S <= 0 || C <= 0 || C > S || ld < S) return TDR_ERR_BAD_ARG;
    if (S > CL_PP * CL_TH) return TDR_ERR_UNSUPPORTED;
    if (g_maxmin_two)
        hipLaunchKernelGGL(cluster_maxmin2_kernel, dim3(1), dim3(CL_TH), 0, (hipStream_t)stream, D2, ld, S, C, seeds, C, 0.f,
                           (int32_t*)nullptr);
    else
        hipLaunchKernelGGL(cluster_maxmin_kernel, dim3(1), dim3(CL_TH), 0, (hipStream_t)stream, D2, ld, S, C, seeds, C, 0.f,
                           (int32_t*)nullptr);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* The same seeding with the number of seeds read off the data: up to c_max farthest-point seeds; the first step t >= c_min
 * at which the max-min (squared) distance falls below `drop` x the previous one ends the loop with t seeds -- the point
 * where every well-separated group holds a seed -- else c_min seeds.  *n_seeds (device int32) receives the count; seeds
 * has room for c_max entries. */
int tdr_cluster_maxmin_adaptive_f32(const float* D2, int64_t ld, int S, int c_min, int c_max, float drop, int32_t* seeds,
                                    int32_t* n_seeds, void* stream) {
    if (!D2 || !seeds || !n_seeds || S <= 0 || c_min <= 0 || c_max < c_min || c_max > S || ld < S || !(drop > 0.f && drop < 1.f))
        return TDR_ERR_BAD_ARG;
    if (S > CL_PP * CL_TH) return TDR_ERR_UNSUPPORTED;
    if (g_maxmin_two)
        hipLaunchKernelGGL(cluster_maxmin2_kernel, dim3(1), dim3(CL_TH), 0, (hipStream_t)stream, D2, ld, S, c_max, seeds, c_min, drop,
                           n_seeds);
    else
        hipLaunchKernelGGL(cluster_maxmin_kernel, dim3(1), dim3(CL_TH), 0, (hipStream_t)stream, D2, ld, S, c_max, seeds, c_min, drop,
                           n_seeds);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Per-tile lower bounds of the distance to every cluster centre, from a block of exact squared distances:
 * d2 (rows, C; row stride ld) = squared distances of `rows` consecutive rows of the padded cluster-sorted order (rows a multiple
 * of 32) to the C centres, as the dense kernel gives them (norm expansion, fp32); row_map (rows): source row or -1 (padding,
 * ignored); xn (rows) / cn (C): squared norms of the rows / centres.  out (rows / 32, C): for every tile and centre
 * sqrt(max(0, min over its valid rows of d2 - eps (|x|^2 + |c|^2))), rounded down -- eps = (d + 16) 2^-23 covers the rounding
 * of the expansion, so the value never exceeds the true distance of any row of the tile; tiles of padding only get +inf. */
int tdr_cluster_tile_cdist_f32(const float* d2, int64_t ld, int64_t rows, int C, int d, const int32_t* row_map, const float* xn,
                               const float* cn, float* out, void* stream) {
    if (!d2 || !row_map || !xn || !cn || !out || rows <= 0 || rows % 32 != 0 || C <= 0 || ld < C || d <= 0) return TDR_ERR_BAD_ARG;
    const dim3 grid((unsigned)(rows / 32), (unsigned)((C + 255) / 256));
    hipLaunchKernelGGL(tile_cdist_kernel, grid, dim3(256), 0, (hipStream_t)stream, d2, ld, C, (float)(d + 16) * 1.1920929e-07f, row_map, xn, cn, out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* out (m, d) = X[idx[idx2 ? idx2[i] : i]] */
int tdr_gather_rows_f32(const float* X, int64_t ldx, int d, const int32_t* idx, const int32_t* idx2, int64_t m, float* out,
                        void* stream) {
    if (!X || !idx || !out || m <= 0 || d <= 0 || ldx < d) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((m * d + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, ldx, d, idx,
                       idx2, m, out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* 3. one Lloyd update on the sample: cent (C, d) <- mean of the sample points labelled c (clusters without points keep
 * their centre).  ws: C * d floats + C int32. */
int tdr_cluster_update_f32(const float* Xs, int64_t S, int d, const int32_t* labels, int C, float* cent, void* ws, void* stream) {
    if (!Xs || !labels || !cent || S <= 0 || d <= 0 || C <= 0) return TDR_ERR_BAD_ARG;
    (void)ws;  // kept in the signature: the update needs no scratch any more
    hipLaunchKernelGGL(centroid_update_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, Xs, S, d, labels, cent);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

static inline void cluster_runs(int64_t n, int* rps, int* NB) {
    int64_t per = (n + 2047) / 2048;
    per = (per + 63) / 64 * 64;
    if (per < 1024) per = 1024;
    *rps = (int)per;
    *NB = (int)((n + per - 1) / per);
}

/* bytes of the scratch block tdr_cluster_tables_f32 needs */
int64_t tdr_cluster_tables_workspace_bytes(int64_t n, int C) {
    if (n <= 0 || C <= 0) return 0;
    int rps, NB;
    cluster_runs(n, &rps, &NB);
    return ((int64_t)2 * C + 1 + (int64_t)C * NB) * 4;
}

/* 4 + 5. everything that follows the assignment of all n points (labels): radii (rounded up), cluster sizes, the padded
 * cluster-sorted layout (row_map: n + 32 C int32, -1 = padding; tile_cluster: (n + 32 C) / 32 int32; tile_begin: C + 1;
 * tiles: C; *n_img = rows of the padded image), centre distances (C x C, rounded down) and visiting order (C x C).
 * The members of a cluster are laid out by ascending row (deterministic).  perm / inv / ppos (n int32 each, all or none):
 * the same order without padding: position -> row, row -> position, position -> position in the padded layout.  ws: tdr_cluster_tables_workspace_bytes(n, C). */
// fork / join pair of tdr_cluster_tables_f32 (one per device, created on first use, never destroyed): the centre tables depend on
// the centres alone and run beside the radii / histogram / scatter chain instead of behind it (0.39 ms of the headline's kNN build,
// whose critical path is the index build)
static int cluster_side(hipStream_t* side, hipEvent_t* fork, hipEvent_t* join) {
    static hipStream_t g_side[16] = {};
    static hipEvent_t g_fork[16] = {}, g_join[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (!g_side[dev]) {
        hipStream_t s; hipEvent_t a, b;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 0;
        if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) return 0;
        g_side[dev] = s; g_fork[dev] = a; g_join[dev] = b;
    }
    *side = g_side[dev]; *fork = g_fork[dev]; *join = g_join[dev];
    return 1;
}

int tdr_cluster_tables_f32(const float* X, int64_t n, int d, int64_t ldx, const int32_t* labels, const float* cent, int C,
                           float* radius, int32_t* tile_begin, int32_t* tiles, int32_t* tile_cluster, int32_t* row_map,
                           int64_t* n_img, float* dist, int32_t* order, int32_t* perm, int32_t* inv, int32_t* ppos, void* ws,
                           int64_t ws_bytes, void* stream) {
    if (!X || !labels || !cent || !radius || !tile_begin || !tiles || !tile_cluster || !row_map || !n_img || !dist || !order || !ws)
        return TDR_ERR_BAD_ARG;
    if (n <= 0 || d <= 0 || ldx < d || C <= 0 || n >= 0x7fffffffLL || (perm == nullptr) != (inv == nullptr) ||
        (perm == nullptr) != (ppos == nullptr))
        return TDR_ERR_BAD_ARG;
    if (C > 4096) return TDR_ERR_UNSUPPORTED;
    if (ws_bytes < tdr_cluster_tables_workspace_bytes(n, C)) return TDR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int rps, NB;
    cluster_runs(n, &rps, &NB);
    int32_t* counts = (int32_t*)ws;
    int32_t* row_begin = counts + C;
    int32_t* H = row_begin + C + 1;
    // centre distances + visiting orders on the side stream, joined at the end (a capturing stream keeps everything in line)
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    bool forked = false;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone && cluster_side(&side, &ev_fork, &ev_join)) {
        if (hipEventRecord(ev_fork, st) == hipSuccess && hipStreamWaitEvent(side, ev_fork, 0) == hipSuccess) {
            hipLaunchKernelGGL(centre_tables_kernel, dim3((unsigned)C), dim3(256), (size_t)C * sizeof(float), side, cent, C, d, dist, order);
            forked = hipGetLastError() == hipSuccess && hipEventRecord(ev_join, side) == hipSuccess;
            if (!forked) hipStreamSynchronize(side);      // whatever did get enqueued ends before the in-line launch below
        }
    }
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)tdr_cluster_tables_workspace_bytes(n, C), st);
    if (e == hipSuccess) e = hipMemsetAsync(radius, 0, (size_t)C * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(row_map, 0xFF, (size_t)(n + 32 * (int64_t)C) * 4, st);
    if (e != hipSuccess) { if (forked) hipStreamSynchronize(side); return (int)e; }
    hipLaunchKernelGGL(cluster_radius_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, X, n, d, ldx, labels, cent,
                       (unsigned*)radius, counts);
    hipLaunchKernelGGL(radius_round_up_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, radius, C);
    hipLaunchKernelGGL(cluster_tiles_kernel, dim3(1), dim3(256), 0, st, (const int32_t*)counts, C, tile_begin, tiles, tile_cluster, n_img,
                       row_begin);
    hipLaunchKernelGGL(cluster_hist_kernel, dim3((unsigned)((NB + 3) / 4)), dim3(256), 0, st, labels, n, rps, NB, H);
    hipLaunchKernelGGL(cluster_hist_scan_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, st, H, C, NB);
    hipLaunchKernelGGL(cluster_scatter_kernel, dim3((unsigned)((NB + 3) / 4)), dim3(256), 0, st, labels, n, rps, NB,
                       (const int32_t*)tile_begin, (const int32_t*)row_begin, H, row_map, perm, inv, ppos);
    if (forked) {
        if (hipStreamWaitEvent(st, ev_join, 0) != hipSuccess) hipStreamSynchronize(side);
    } else {
        hipLaunchKernelGGL(centre_tables_kernel, dim3((unsigned)C), dim3(256), (size_t)C * sizeof(float), st, cent, C, d, dist, order);
    }
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}


/* Predicted share of the database tiles a pruned scan still visits at threshold tau (squared distance units), from the cluster
 * tables: sum over query clusters w and database clusters c with max(0, dist[w, c] - radius[w] - radius[c])^2 <= tau of
 * tiles[w] * tiles[c], and (sum tiles)^2 -- both as exact integers in out (2 x uint64, caller-zeroed): ONE launch and one host read
 * where the torch formulation of ClusterIndex.scan_fraction took thirteen launches per evaluation (0.75 ms of the 27 ms kNN build
 * went into three of those and their reads).  The same integers on every rank. */
int tdr_cluster_scan_fraction_f32(const float* dist, const float* radius, const int32_t* tiles, int C, float tau, void* out, void* stream) {
    if (!dist || !radius || !tiles || !out || C <= 0) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(scan_fraction_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, dist, radius, tiles, C, tau,
                       (unsigned long long*)out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
