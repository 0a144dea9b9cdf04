// K5p -- UMAP's gradient on the per-iteration firing lists (tdr_umap_sched.hip) with the negatives served from LDS.
//
// Replaces the same reference lines as umap_sched_grad_kernel:
//   neighbor_embedding/umap.py:236-264   attraction over the edges that fire at this iteration (lists of the schedule build)
//   neighbor_embedding/umap.py:266-292   repulsion over min(5 * (#fired edges), n_negatives) sampled negatives
//   neighbor_embedding/base.py:617-649   negative sampling
// What changes is WHERE a negative comes from.  The i.i.d. sampler of umap_sched_grad_kernel issues one random 8-byte L2
// request per negative (43 M per iteration at N = 1M: the launch sits at 0.69 of the L2's gather-request ceiling with the
// vector ALUs right behind, and the embedding has to be cut into L2-sized slices with a second visit of every row per
// slice).  Here a workgroup owns ROWS consecutive rows (a GLOBAL row block: the same partition whatever the row sharding)
// and, per iteration, stages a POOL of RUNS runs of 16 consecutive rows of Z into LDS -- every run chosen uniformly among
// the ceil(N / 16) runs by a counter hash of (seed, iteration, block, slot), i.e. RUNS independent uniform locations of the
// data set, each one 128-byte line of Z (nc = 2) fetched with one coalesced request.  A row then draws each of its
// negatives uniformly among the 16 RUNS pool rows (hash of (seed, iteration, row, item)) with one LDS read:
//   * marginal law: every row of Z is drawn with probability 1 / (16 ceil(N / 16)) per item -- uniform.  A draw that lands on
//     the row itself or on the padding of the last run contributes exactly zero force (self: z_i - z_j = 0; padding: a
//     sentinel at 1e30 makes the coefficient 0), i.e. it is DROPPED, where the reference re-maps self (r >= i -> r + 1):
//     conditional on being kept a draw is uniform over the other N - 1 rows, and the number kept is n - Binomial(n, <= 16/N);
//   * joint law: the items of one row are independent given the pool; the rows of one block share the pool of an iteration
//     (two rows pick the same negative with probability 1 / (16 RUNS) instead of 1 / N), and pools of different blocks /
//     iterations are independent.  tests/test_umap_pool_gpu.py: chi-square of the marginal, run uniformity, independence of
//     consecutive iterations; the embedding quality gates of tests/golden/quality.json.
// L2 requests per iteration at N = 1M: 1M (pool lines) + 8.6 M (fired-edge gathers) instead of 52 M, no slices, one visit
// per row whatever N, and no cross-lane reduction: ONE LANE PER ROW (the row's sums stay in registers; the order of a row's
// sum is its list order, then its item order -- a function of the row alone, so a row-sharded fit reproduces the
// single-process fit bit for bit).  Lanes of a wavefront run as long as the busiest row: the rows of a block are
// counting-sorted by active count in LDS first (a row has 5 negatives per fired edge, so one key serves both loops).
#include "tdr_embed_common.h"
#include "tdr_umap_pool.h"
#include "../../include/torchdr_amd.h"

namespace tdr {

typedef __attribute__((address_space(1))) const void* pgptr_t;
typedef __attribute__((address_space(3))) void* plptr_t;

constexpr float POOL_SENTINEL = 1e30f;   // coordinates of the padding rows of the last run: d = inf, coefficient 0

// ---- the sampler (shared by the gradient kernel and the debug dump) ----------------------------------------------------
__device__ __forceinline__ uint32_t pool_block_key(uint64_t seed, uint32_t iter, uint32_t gb) {
    uint32_t h = mix32(gb ^ (uint32_t)(seed >> 32) ^ 0x5bd1e995u);
    h = mix32(h + (uint32_t)seed);
    return mix32(h ^ (iter * 0x85EBCA6Bu + 0x27D4EB2Fu));
}
// run of Z held by pool slot `slot` (uniform over the n_runs = ceil(N / 16) runs)
__device__ __forceinline__ uint32_t pool_run(uint32_t bkey, uint32_t slot, uint32_t n_runs) {
    return __umulhi(mix32_item(bkey + slot * 0x9E3779B9u), n_runs);
}
__device__ __forceinline__ uint32_t pool_row_key(uint64_t seed, uint32_t iter, int64_t gi) {
    return neg_row_key(seed, iter, gi) ^ 0x68E31DA4u;    // decorrelated from the i.i.d. sampler's stream of the same row
}
// pool row (0 .. 16 RUNS - 1) of item k of a row
template <int LOGP>
__device__ __forceinline__ uint32_t pool_item(uint32_t rkey, uint32_t k) {
    return mix32_item(rkey + k * 0x9E3779B9u) >> (32 - LOGP);
}

constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }

template <int NC>
__device__ __forceinline__ Vec<NC> pool_read(const float* pool, uint32_t s) {
    Vec<NC> r;
    if (NC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(pool + s * 2);
        r.v[0] = t.x; r.v[1] = t.y;
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) r.v[c] = pool[s * NC + c];
    }
    return r;
}

// ROWS threads = ROWS rows of a global row block; RUNS pool runs of 16 rows
template <int NC, int ROWS, int RUNS>
__global__ __launch_bounds__(ROWS) void umap_pool_grad_kernel(const PoolGradParams P) {
    constexpr int POOL_ROWS = RUNS * 16;
    constexpr int LOGP = ilog2(POOL_ROWS);
    static_assert((1 << LOGP) == POOL_ROWS, "pool rows must be a power of two");
    constexpr int PPR = 4 * NC;                 // 16-byte pieces per run
    constexpr int PIECES = RUNS * PPR;
    constexpr int NW = ROWS / 64;
    static_assert(PIECES % ROWS == 0, "pool pieces must divide over the threads");
    constexpr int NIT = PIECES / ROWS;
    constexpr int U = 4;
    __shared__ __attribute__((aligned(16))) float pool[POOL_ROWS * NC];
    __shared__ uint32_t hist[64];
    __shared__ uint16_t order[ROWS];
    __shared__ uint2 rec[ROWS];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t iter = P.iter + (P.iter_base ? (uint32_t)*P.iter_base : 0u);
    const int64_t gb = P.gb0 + (int64_t)blockIdx.x;
    const uint32_t bkey = pool_block_key(P.seed, iter, (uint32_t)gb);

    // 1. the pool: NIT LDS-DMA instructions per wavefront (a lane moves 16 bytes; the 4 NC lanes of a run one 64 NC-byte run)
    const bool ragged = (P.n_total & 15) != 0;
    const int64_t z_floats = P.n_total * NC;
    uint32_t fix = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = (it * NW + wave) * 64 + lane;
        const uint32_t slot = (uint32_t)p / PPR, part = (uint32_t)p % PPR;
        const uint32_t run = pool_run(bkey, slot, P.n_runs);
        const int64_t off = (int64_t)run * (16 * NC) + part * 4;
        const bool last = ragged && run == P.n_runs - 1u;
        if (last) fix |= 1u << it;
        const float* src = (last && off + 4 > z_floats) ? P.Z : P.Z + off;
        __builtin_amdgcn_global_load_lds((pgptr_t)src, (plptr_t)(pool + (it * NW + wave) * 256), 16, 0, 0);
    }
    // 2. this thread's row and its record of the iteration
    if (t < 64) hist[t] = 0;
    const int64_t gi0 = gb * ROWS + t;
    const int64_t r0 = gi0 - P.row0;
    uint2 h = make_uint2(0u, 0u);
    if (r0 >= 0 && r0 < P.n_rows) {
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t hv = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(P.hdr + (size_t)P.t_local * P.n_rows + r0));
        h = make_uint2(hv.x, hv.y);
    }
    const uint32_t act0 = h.y >> 16;
    const uint32_t key = 63u - (act0 < 63u ? act0 : 63u);     // busiest rows first
    __syncthreads();   // pool staged (the barrier waits for the wavefront's DMA), histogram zeroed
    if (fix) {
        // padding rows of the last run (N % 16 != 0): the pieces of that run are rewritten float by float, rows >= N as sentinels
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (!(fix >> it & 1u)) continue;
            const int p = (it * NW + wave) * 64 + lane;
            const uint32_t part = (uint32_t)p % PPR;
            const int64_t off = (int64_t)(P.n_runs - 1u) * (16 * NC) + part * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) pool[p * 4 + e] = off + e < z_floats ? P.Z[off + e] : POOL_SENTINEL;
        }
    }
    const uint32_t rank = atomicAdd(&hist[key], 1u);
    __syncthreads();
    if (t < 64) {   // exclusive scan of the 64 bins
        const uint32_t v = hist[t];
        uint32_t s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(s, o, 64);
            if (lane >= o) s += u;
        }
        hist[t] = s - v;
    }
    __syncthreads();
    order[hist[key] + rank] = (uint16_t)t;
    rec[t] = h;
    __syncthreads();
    // 3. the row this lane evaluates (position t of the sorted order)
    const int my = order[t];
    h = rec[my];
    const int64_t gi64 = gb * ROWS + my;
    const int64_t r = gi64 - P.row0;
    if (r < 0 || r >= P.n_rows) return;
    const uint32_t gi = (uint32_t)gi64;
    const Vec<NC> zi = load_z<NC>(P.Z, gi64);
    const int npos = (int)(h.y & 0xffffu);
    int n_use = (int)(h.y >> 16) * P.neg_rate;
    if (n_use > P.n_negatives) n_use = P.n_negatives;
    const float two_ab = 2.0f * P.a * P.b, m2b = -2.0f * P.b;
    float ga[NC], gr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ga[c] = 0.f; gr[c] = 0.f; }
    // attraction: the row's fired edges, four list entries per 16-byte read (the list carries 64 entries of slack)
    const int32_t* lst = P.list + h.x;
    for (int k = 0; k < npos; k += U) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4 l4;
        __builtin_memcpy(&l4, lst + k, 16);
        Vec<NC> zj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) zj[u] = load_z<NC>(P.Z, (int64_t)(k + u < npos ? (uint32_t)l4[u] : gi));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float df[NC];
            const float d = sqdist<NC>(zi, zj[u], df);
            const float pb = fast_pow(d, P.b);
            // 2ab d^(b-1) / (1 + a d^b), 0 where d <= 0 (umap.py:252-256)
            float coef = pb * two_ab * fast_rcp(d * (1.0f + P.a * pb));
            if (!(k + u < npos) || !(d > 0.f)) coef = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) ga[c] += coef * df[c];
        }
    }
    // repulsion: n_use items from the pool; -2b / ((d + eps)(1 + a d^b)) (umap.py:272-281)
    const uint32_t rkey = pool_row_key(P.seed, iter, gi64);
    for (int k = 0; k < n_use; k += U) {
        Vec<NC> zj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) zj[u] = pool_read<NC>(pool, pool_item<LOGP>(rkey, (uint32_t)(k + u)));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float df[NC];
            const float d = sqdist<NC>(zi, zj[u], df);
            const float pb = fast_pow(d, P.b);
            float coef = m2b * fast_rcp((d + P.eps) * (1.0f + P.a * pb));
            if (!(k + u < n_use)) coef = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) gr[c] += coef * df[c];
        }
    }
    float g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = P.exag * fminf(fmaxf(ga[c], -4.f), 4.f) + P.rep * fminf(fmaxf(gr[c], -4.f), 4.f);
    if (NC == 2) {
        *reinterpret_cast<float2*>(P.grad + (size_t)r * 2) = make_float2(g[0], g[1]);
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) P.grad[(size_t)r * NC + c] = g[c];
    }
}

// test hook: the global row of every item the gradient kernel draws for rows with nuse[r] items (-1: beyond the row's count;
// -2: a dropped draw -- the row itself or the padding of the last run)
template <int ROWS, int RUNS>
__global__ __launch_bounds__(256) void umap_pool_debug_kernel(uint64_t seed, uint32_t iter, int64_t n_total, int64_t row0, int64_t n_rows,
                                                              const int32_t* __restrict__ nuse, int width, int64_t* __restrict__ out) {
    constexpr int LOGP = ilog2(RUNS * 16);
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t gi = row0 + r;
    const uint32_t n_runs = (uint32_t)((n_total + 15) / 16);
    const uint32_t bkey = pool_block_key(seed, iter, (uint32_t)(gi / ROWS));
    const uint32_t rkey = pool_row_key(seed, iter, gi);
    const int n = nuse[r];
    for (int k = 0; k < width; ++k) {
        int64_t j = -1;
        if (k < n) {
            const uint32_t s = pool_item<LOGP>(rkey, (uint32_t)k);
            j = (int64_t)pool_run(bkey, s >> 4, n_runs) * 16 + (s & 15u);
            if (j >= n_total || j == gi) j = -2;
        }
        out[(size_t)r * width + k] = j;
    }
}

template <int NC, int ROWS, int RUNS>
static int launch_pool(const PoolGradParams& P0, hipStream_t st) {
    PoolGradParams P = P0;
    P.gb0 = P.row0 / ROWS;
    const int64_t gb1 = (P.row0 + P.n_rows - 1) / ROWS;
    hipLaunchKernelGGL((umap_pool_grad_kernel<NC, ROWS, RUNS>), dim3((unsigned)(gb1 - P.gb0 + 1)), dim3(ROWS), 0, st, P);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}

template <int NC>
static int launch_pool_geom(const PoolGradParams& P, int geom, hipStream_t st) {
    switch (geom) {
        case 1: return launch_pool<NC, 256, 256>(P, st);
        case 2: return launch_pool<NC, 512, 256>(P, st);
        case 3: return launch_pool<NC, 512, 512>(P, st);
        case 4: return launch_pool<NC, 1024, 256>(P, st);
        case 5: return launch_pool<NC, 1024, 512>(P, st);
        default: return launch_pool<NC, TDR_POOL_ROWS, TDR_POOL_RUNS>(P, st);
    }
}

int launch_pool_grad(const PoolGradParams& P, int geom, hipStream_t st) {
    if (P.nc == 2) return launch_pool_geom<2>(P, geom, st);
    if (P.nc == 3) return launch_pool_geom<3>(P, geom, st);
    return TDR_ERR_UNSUPPORTED;
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* 1 when the pool-sampled gradient kernel serves embeddings of nc components. */
int tdr_umap_pool_supported(int nc) { return nc == 2 || nc == 3; }

/* tdr_umap_sched_grad_f32 (n_slices = 1 lists) with the negatives drawn from a per-block LDS pool (file header). */
int tdr_umap_pool_grad_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* list, const void* hdr,
                           int t_local, float a, float b, int n_iter, int neg_rate, int n_negatives, uint64_t seed, float exag,
                           float rep, float eps, float* grad, int geom, void* stream) {
    if (!Z || !list || !hdr || !grad || n_rows <= 0 || row0 < 0 || n_total < 2 || n_total >= 0x7fffffffLL) return TDR_ERR_BAD_ARG;
    if (t_local < 0 || t_local >= 32 || neg_rate < 0 || n_negatives < 0 || geom < 0 || geom > 5) return TDR_ERR_BAD_ARG;
    if (((uintptr_t)Z & 15u) != 0) return TDR_ERR_BAD_ARG;
    if (!tdr_umap_pool_supported(nc)) return TDR_ERR_UNSUPPORTED;
    PoolGradParams P = {};
    P.Z = Z; P.nc = nc; P.n_total = n_total; P.row0 = row0; P.n_rows = n_rows; P.list = list; P.hdr = (const uint2*)hdr;
    P.t_local = t_local; P.a = a; P.b = b; P.neg_rate = neg_rate; P.n_negatives = n_negatives; P.seed = seed; P.iter = (uint32_t)n_iter;
    P.iter_base = nullptr; P.exag = exag; P.rep = rep; P.eps = eps; P.grad = grad; P.n_runs = (uint32_t)((n_total + 15) / 16);
    return launch_pool_grad(P, geom, (hipStream_t)stream);
}

int tdr_umap_pool_debug_negatives(uint64_t seed, int n_iter, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nuse, int geom,
                                  int width, int64_t* out, void* stream) {
    if (!nuse || !out || n_rows <= 0 || width <= 0 || n_total < 2 || geom < 0 || geom > 5) return TDR_ERR_BAD_ARG;
    const dim3 grid((unsigned)((n_rows + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
#define TDR_POOL_DBG(R, Q) hipLaunchKernelGGL((umap_pool_debug_kernel<R, Q>), grid, dim3(256), 0, st, seed, (uint32_t)n_iter, n_total, row0, n_rows, nuse, width, out)
    switch (geom) {
        case 1: TDR_POOL_DBG(256, 256); break;
        case 2: TDR_POOL_DBG(512, 256); break;
        case 3: TDR_POOL_DBG(512, 512); break;
        case 4: TDR_POOL_DBG(1024, 256); break;
        case 5: TDR_POOL_DBG(1024, 512); break;
        default: TDR_POOL_DBG(TDR_POOL_ROWS, TDR_POOL_RUNS); break;
    }
#undef TDR_POOL_DBG
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
