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
// A row's items walk the pool in a per-row arithmetic progression: item k reads pool row (alpha + k beta) >> (32 - LOGP) with
// alpha, beta (odd) hashed from (seed, iteration, row).  Every item is uniform over the pool (alpha is), any two items of a row
// are independent (their difference (k' - k) beta is uniform), the items of a row never coincide while it has fewer items than
// the pool has rows, and two rows share a progression with probability 2^-31 -- for ONE add per item where a strong hash of the
// item index costs two 32-bit multiplies (quarter rate: a third of the instruction cycles of an item).
struct PoolRowKey { uint32_t alpha, beta; };
__device__ __forceinline__ PoolRowKey pool_row_key(uint64_t seed, uint32_t iter, int64_t gi) {
    PoolRowKey K;
    K.alpha = neg_row_key(seed, iter, gi) ^ 0x68E31DA4u;    // decorrelated from the i.i.d. sampler's stream of the same row
    K.beta = mix32_item(K.alpha + 0x632BE5ABu) | 1u;
    return K;
}
// pool row (0 .. POOL_ROWS - 1) of item k of a row
template <int LOGP>
__device__ __forceinline__ uint32_t pool_item(const PoolRowKey& K, uint32_t k) {
    return (K.alpha + k * K.beta) >> (32 - LOGP);
}

constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }

// squared distance with fused multiply-adds (this kernel is compared with the oracle at 1e-5, not bit for bit with the other
// gradient kernels; the library is built with -ffp-contract=off, so the fusion is explicit)
template <int NC>
__device__ __forceinline__ float sqdist_fma(const Vec<NC>& a, const Vec<NC>& b, float (&df)[NC]) {
    df[0] = a.v[0] - b.v[0];
    float d = df[0] * df[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) { df[c] = a.v[c] - b.v[c]; d = __builtin_fmaf(df[c], df[c], d); }
    return d;
}

// Row gather through a buffer descriptor: a lane whose byte offset lies beyond the descriptor's range gets zeros and issues NO
// request -- the branch-free way to leave lanes out of a gather (the L1 serves divergent 8-byte gathers at ~0.44 lanes per clock
// and CU, tools/gather_bench.hip: what bounds the attraction phase is the number of lanes that ask)
typedef int pool_i32x2 __attribute__((ext_vector_type(2)));
typedef int pool_i32x3 __attribute__((ext_vector_type(3)));
template <int NC>
__device__ __forceinline__ Vec<NC> gather_z(__amdgpu_buffer_rsrc_t rs, uint32_t voff) {
    Vec<NC> r;
    if (NC == 2) {
        const pool_i32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, 0, 0);
        r.v[0] = __int_as_float(t.x); r.v[1] = __int_as_float(t.y);
    } else {
        const pool_i32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)voff, 0, 0);
        r.v[0] = __int_as_float(t.x); r.v[1] = __int_as_float(t.y); r.v[NC - 1] = __int_as_float(t.z);
    }
    return r;
}

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

// parts of a row that fires `act` edges, and the end of part j's share of `n` units (entries or groups of negatives)
constexpr uint32_t POOL_CH = 16, POOL_MMAX = 8;
__device__ __forceinline__ uint32_t pool_parts(uint32_t act) {
    if (act <= POOL_CH) return 1u;
    const uint32_t m = (act + POOL_CH - 1u) / POOL_CH;
    return m < POOL_MMAX ? m : POOL_MMAX;
}
__device__ __forceinline__ uint32_t pool_part_end(uint32_t n, uint32_t j, uint32_t m) { return n * j / m; }

// the row's gradient (clamps of umap.py:262,290) and, when the launch carries the step (P.Z_out), torch.optim.SGD's update of the
// row written to the OTHER embedding buffer (every row of this launch still reads the old one) with check_NaNs' flag
// (affinity_matcher.py:315,427)
// In the loop object (P.lr_table) the learning rate comes from the device table at the iteration, and at the iterations the
// reference inspects (iteration % check_interval == 0, affinity_matcher.py:331-349) the row also leaves its gradient, its stepped
// position in `snap` and its share of the squared gradient norm (returned; the caller adds the block's shares to norm2) -- what
// sgd_table_step_kernel does for the unfused sequence.
template <int NC>
__device__ __forceinline__ float pool_store_grad(const PoolGradParams& P, int64_t r, int64_t gi, const Vec<NC>& zi, uint32_t iter,
                                                 bool inspected, const float (&ga)[NC], const float (&gr)[NC]) {
    float g[NC];
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = P.exag * fminf(fmaxf(ga[c], -4.f), 4.f) + P.rep * fminf(fmaxf(gr[c], -4.f), 4.f);
    if (inspected) {
#pragma unroll
        for (int c = 0; c < NC; ++c) q += g[c] * g[c];
    }
    if (P.grad && (inspected || !P.lr_table)) {
        if (NC == 2) {
            *reinterpret_cast<float2*>(P.grad + (size_t)r * 2) = make_float2(g[0], g[1]);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) P.grad[(size_t)r * NC + c] = g[c];
        }
    }
    if (P.Z_out) {
        float z[NC];
        bool nan = false;
        const float lr = P.lr_table ? P.lr_table[iter] : P.lr;
#pragma unroll
        for (int c = 0; c < NC; ++c) { z[c] = __builtin_fmaf(-lr, g[c], zi.v[c]); nan = nan || z[c] != z[c]; }
        if (NC == 2) {
            *reinterpret_cast<float2*>(P.Z_out + (size_t)gi * 2) = make_float2(z[0], z[1]);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) P.Z_out[(size_t)gi * NC + c] = z[c];
        }
        if (inspected && P.snap) {
#pragma unroll
            for (int c = 0; c < NC; ++c) P.snap[(size_t)r * NC + c] = z[c];
        }
        if (nan) atomicCAS(P.nan_flag, 0, (int)iter + 1);
    }
    return q;
}

// THREADS lanes evaluate the ROWS = RPT x THREADS rows of a global row block against a pool of RUNS runs of RUNLEN rows.
// RPT = 2 ("folded"): position p of the sorted order and position ROWS - 1 - p go to the same lane, one after the other -- a
// wavefront then carries a busy and a quiet batch and all wavefronts of the block end together (sorted but unfolded, the first
// wavefront holds the 64 busiest rows, runs ~2.7x the average and keeps the block's LDS allocated while the others idle).
// DBG instances (tools/umap_pool_perf.py only) honour the ablation switches of P.ablate and write phase time stamps of every
// block's first wavefront to P.dbg_times; the production instance carries neither.
// (An LDS window of the rows around the block for the fired-edge gathers -- in the loop's cluster-sorted numbering 42 % of the
// fired edges end in the row's own block, all within 1024 rows -- was built in three forms and measured slower every time:
// profiles/r06_pool_window.json.)
// SPLIT > 1: the global row block (and its pool) stays BROWS = SPLIT x ROWS rows, but SPLIT workgroups share it, each staging the
// whole pool and evaluating ROWS of its rows -- a row's sums are a function of the row alone, so the bits are those of the unsplit
// launch; what changes is the number of wavefronts and the rows a lane evaluates one after the other: a rank of an 8-rank fit at
// N = 1M holds 123 blocks of 1024 rows = 984 wavefronts of two rows per lane, one per SIMD, 21.5 us per launch where an eighth of
// the single-process launch would be 7.
template <int NC, int THREADS, int RPT, int RUNS, int RUNLEN, int SPLIT, bool DBG>
__global__ __launch_bounds__(THREADS, NC == 2 ? 6 : 4) void umap_pool_grad_kernel(const PoolGradParams P) {   // 2 components: <= 80 registers (capping at 64 spills ten and measured slower)
    const int ablate = DBG ? P.ablate : 0;
    auto stamp = [&](int i) {
        if (DBG && P.dbg_times && (threadIdx.x & 63) == 0)
            P.dbg_times[((size_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)) * 8 + i] = (unsigned long long)__builtin_amdgcn_s_memtime();
    };
    stamp(0);
    constexpr int ROWS = THREADS * RPT;          // rows of this workgroup
    constexpr int BROWS = ROWS * SPLIT;          // rows of the global block (the unit the pool is keyed by)
    constexpr int POOL_ROWS = RUNS * RUNLEN;
    constexpr int LOGP = ilog2(POOL_ROWS), LOGR = ilog2(RUNLEN);
    static_assert((1 << LOGP) == POOL_ROWS && (1 << LOGR) == RUNLEN, "pool rows / run length must be powers of two");
    constexpr int PPR = RUNLEN * NC / 4;        // 16-byte pieces per run
    static_assert(PPR * 4 == RUNLEN * NC, "a run must be made of whole 16-byte pieces");
    constexpr int PIECES = RUNS * PPR;
    constexpr int NW = THREADS / 64;
    static_assert(PIECES % THREADS == 0, "pool pieces must divide over the threads");
    constexpr int NIT = PIECES / THREADS;
    constexpr int U = 4;
    // PARTS: a row that fires more than POOL_CH edges is cut into m = ceil(act / POOL_CH) <= POOL_MMAX parts (even shares of its
    // listed edges and of its groups of negatives) that different lanes evaluate; the parts' sums meet in LDS and are added in part
    // order.  Without it the lane that holds a hub row (a few rows per block fire 40-150 edges where the mean is 8.6) runs 5-15x as
    // long as its neighbours and its wavefront decides when the block ends (profiles/r06_pool_phases.json: the first wavefront of the
    // sorted order took 2x the others').  m is a function of the row alone and the parts are added in order whoever evaluates them
    // (a row whose parts do not fit the block's XCAP slots is evaluated by ONE lane part by part): the bits of a row do not depend on
    // the block, the launch or the sharding.
    constexpr int XCAP = 384;                   // part slots of split rows per block
    constexpr int ITEMS = ROWS + XCAP;          // base items (one per row: part 0, or the whole row) + extra items (parts >= 1)
    constexpr int NPASS = (ITEMS + THREADS - 1) / THREADS;
    __shared__ __attribute__((aligned(16))) float pool[POOL_ROWS * NC];
    __shared__ uint32_t hist[64];
    __shared__ uint32_t xcount;                 // part slots handed out
    __shared__ uint32_t scount;                 // split rows
    __shared__ uint16_t order[ITEMS];
    __shared__ uint2 rec[ROWS];
    __shared__ uint32_t rowx[ROWS];             // parts m (low 8 bits) | first part slot + 1 << 8 (0: not split)
    __shared__ uint32_t xitem[XCAP];            // extra item e: row | part << 16
    __shared__ uint32_t srow[XCAP / 2];         // split rows
    __shared__ float psum[XCAP * 2 * NC];       // (attraction, repulsion) sums of the parts of split rows

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t iter = P.iter + (P.iter_base ? (uint32_t)*P.iter_base : 0u);
    const int64_t gb = P.gb0 + (int64_t)(blockIdx.x / SPLIT);
    const int64_t rowbase = gb * BROWS + (int64_t)(blockIdx.x % SPLIT) * ROWS;      // first (global) row of this workgroup
    if (SPLIT > 1 && (rowbase + ROWS <= P.row0 || rowbase >= P.row0 + P.n_rows)) return;   // none of the launch's rows here
    const bool inspected = P.lr_table && P.check_interval > 0 && iter % (uint32_t)P.check_interval == 0u;
    float q2 = 0.f;      // this thread's share of the squared gradient norm (inspected iterations)
    const uint32_t bkey = pool_block_key(P.seed, iter, (uint32_t)gb);

    // 1. the pool: NIT LDS-DMA instructions per wavefront (a lane moves 16 bytes; the PPR lanes of a run one whole run)
    const bool ragged = (P.n_total & (RUNLEN - 1)) != 0;
    const int64_t z_floats = P.n_total * NC;
    uint32_t fix = 0;
    if (!(ablate & 1))
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = (it * NW + wave) * 64 + lane;
        const uint32_t slot = (uint32_t)p / PPR, part = (uint32_t)p % PPR;
        const uint32_t run = pool_run(bkey, slot, P.n_runs);
        const int64_t off = (int64_t)run * (RUNLEN * NC) + part * 4;
        const bool last = ragged && run == P.n_runs - 1u;
        if (last) fix |= 1u << it;
        const float* src = (last && off + 4 > z_floats) ? P.Z : P.Z + off;
        __builtin_amdgcn_global_load_lds((pgptr_t)src, (plptr_t)(pool + (it * NW + wave) * 256), 16, 0, 0);
    }
    // 2. the records of this thread's RPT rows of the block (rows t, t + THREADS, ...), their parts and the sort keys of their items
    if (t < 64) hist[t] = 0;
    if (t == 0) { xcount = 0; scount = 0; }
    uint2 h[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int64_t r0 = rowbase + q * THREADS + t - P.row0;
        h[q] = make_uint2(0u, 0u);
        if (r0 >= 0 && r0 < P.n_rows) {
            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t hv = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(P.hdr + (size_t)P.t_local * P.n_rows + r0));
            h[q] = make_uint2(hv.x, hv.y);
        }
    }
    stamp(1);
    __syncthreads();   // pool staged (the barrier waits for the wavefront's DMA), counters zeroed
    stamp(2);
    if (fix) {
        // padding rows of the last run (N % RUNLEN != 0): the pieces of that run are rewritten float by float, rows >= N as sentinels
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (!(fix >> it & 1u)) continue;
            const int p = (it * NW + wave) * 64 + lane;
            const uint32_t part = (uint32_t)p % PPR;
            const int64_t off = (int64_t)(P.n_runs - 1u) * (RUNLEN * NC) + part * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) pool[p * 4 + e] = off + e < z_floats ? P.Z[off + e] : POOL_SENTINEL;
        }
    }
    uint32_t key[RPT], rank[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int row = q * THREADS + t;
        const uint32_t act = (ablate & 8) ? 0u : h[q].y >> 16;      // one slice: the row's listed edges = its fired edges
        const uint32_t m = pool_parts(act);
        uint32_t base = 0;                                          // first part slot + 1
        if (m > 1 && !(ablate & 64)) {
            const uint32_t b0 = atomicAdd(&xcount, m);
            if (b0 + m <= (uint32_t)XCAP) {
                base = b0 + 1u;
                srow[atomicAdd(&scount, 1u)] = (uint32_t)row;
                for (uint32_t j = 1; j < m; ++j) xitem[b0 + j] = (uint32_t)row | j << 16;    // slot b0 itself is part 0: the row's base item
            }
        }
        rowx[row] = m | base << 8;
        rec[row] = h[q];
        const uint32_t work = base ? pool_part_end(act, 1u, m) : act;   // entries of the base item
        key[q] = 63u - (work < 63u ? work : 63u);                        // busiest items first
        rank[q] = atomicAdd(&hist[key[q]], 1u);
    }
    __syncthreads();
    // extra items = the part slots >= 1 of the split rows.  A slot is an item only if a split row owns it as part >= 1 (part-0
    // slots belong to base items; a refused request advanced the counter without owning anything): the split rows flag theirs
    const uint32_t n_slots = xcount < (uint32_t)XCAP ? xcount : (uint32_t)XCAP;
    constexpr int XPT = (XCAP + THREADS - 1) / THREADS;      // slots a thread looks at
    uint32_t xkey[XPT], xrank[XPT];
    bool have_x[XPT];
    __shared__ uint32_t xvalid[(XCAP + 31) / 32];
    if (t < (XCAP + 31) / 32) xvalid[t] = 0;
    __syncthreads();
    for (uint32_t sidx = (uint32_t)t; sidx < scount; sidx += THREADS) {
        const uint32_t row = srow[sidx];
        const uint32_t rx = rowx[row], m = rx & 255u, b0 = (rx >> 8) - 1u;
        for (uint32_t j = 1; j < m; ++j) atomicOr(&xvalid[(b0 + j) >> 5], 1u << ((b0 + j) & 31u));
    }
    __syncthreads();
#pragma unroll
    for (int xq = 0; xq < XPT; ++xq) {
        const uint32_t e = (uint32_t)(xq * THREADS + t);
        have_x[xq] = e < n_slots && (xvalid[e >> 5] >> (e & 31u) & 1u);
        xkey[xq] = 0; xrank[xq] = 0;
        if (have_x[xq]) {
            const uint32_t xi = xitem[e];
            const uint32_t row = xi & 0xffffu, j = xi >> 16;
            const uint32_t act = rec[row].y >> 16, m = rowx[row] & 255u;
            const uint32_t work = pool_part_end(act, j + 1u, m) - pool_part_end(act, j, m);
            xkey[xq] = 63u - (work < 63u ? work : 63u);
            xrank[xq] = atomicAdd(&hist[xkey[xq]], 1u);
        }
    }
    __syncthreads();
    if (t < 64) {   // exclusive scan of the 64 bins
        const uint32_t v = hist[t];
        uint32_t s2 = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(s2, o, 64);
            if (lane >= o) s2 += u;
        }
        hist[t] = s2 - v;
        if (t == 63) xcount = s2;      // total number of items (reuses the counter)
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPT; ++q) order[hist[key[q]] + rank[q]] = (uint16_t)(q * THREADS + t);
#pragma unroll
    for (int xq = 0; xq < XPT; ++xq)
        if (have_x[xq]) order[hist[xkey[xq]] + xrank[xq]] = (uint16_t)(ROWS + xq * THREADS + t);
    __syncthreads();
    stamp(3);
    const int n_items = (int)xcount;
    const float two_ab = 2.0f * P.a * P.b, m2b = -2.0f * P.b;
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.Z), 0, (int)(uint32_t)(z_floats * 4), 0x00020000);
    // 3. the items this lane evaluates: positions t, 2 THREADS - 1 - t, 2 THREADS + t, ... of the sorted order (busy and quiet
    //    batches alternate per wavefront)
#pragma unroll 1
    for (int q = 0; q < NPASS; ++q) {
        const int pos = (q & 1) ? (q + 1) * THREADS - 1 - t : q * THREADS + t;
        if (pos >= n_items) continue;
        const int it = order[pos];
        int my, j0, j1, slot0 = -1;          // row, parts [j0, j1), first part slot of a split row
        uint32_t m;
        if (it < ROWS) {
            my = it;
            const uint32_t rx = rowx[my];
            m = rx & 255u;
            j0 = 0;
            if (rx >> 8) { j1 = 1; slot0 = (int)(rx >> 8) - 1; } else j1 = (int)m;
        } else {
            const uint32_t xi = xitem[it - ROWS];
            my = (int)(xi & 0xffffu);
            const uint32_t rx = rowx[my];
            m = rx & 255u;
            j0 = (int)(xi >> 16); j1 = j0 + 1; slot0 = (int)(rx >> 8) - 1;
        }
        const uint2 hh = rec[my];
        const int64_t gi64 = rowbase + my;
        const int64_t r = gi64 - P.row0;
        if (r < 0 || r >= P.n_rows) continue;
        const uint32_t gi = (uint32_t)gi64;
        const Vec<NC> zi = load_z<NC>(P.Z, gi64);
        const uint32_t act = hh.y >> 16;
        const int npos_row = (ablate & 2) ? 0 : (int)(hh.y & 0xffffu);
        int n_use = (int)act * P.neg_rate;
        if (n_use > P.n_negatives) n_use = P.n_negatives;
        if (ablate & 4) n_use = 0;
        const int n_grp = P.exact5 ? n_use / 5 : n_use;          // units the negatives are shared out in
        const PoolRowKey rk = pool_row_key(P.seed, iter, gi64);
        const int32_t* lst = P.list + hh.x;
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const bool ab_list = ablate & 32, ab_gather = ablate & 16;      // DBG instances: no list reads / no gathers
        float ga[NC], gr[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { ga[c] = 0.f; gr[c] = 0.f; }
#pragma unroll 1
        for (int j = j0; j < j1; ++j) {
            // part j: listed edges [e0, e1), negatives [kn, kn1)
            const int e0 = (int)pool_part_end((uint32_t)npos_row, (uint32_t)j, m), e1 = (int)pool_part_end((uint32_t)npos_row, (uint32_t)j + 1u, m);
            int kn = (int)pool_part_end((uint32_t)n_grp, (uint32_t)j, m) * (P.exact5 ? 5 : 1);
            const int kn1 = (int)pool_part_end((uint32_t)n_grp, (uint32_t)j + 1u, m) * (P.exact5 ? 5 : 1);
            uint32_t x = rk.alpha + (uint32_t)kn * rk.beta;
            float pa[NC], pr[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) { pa[c] = 0.f; pr[c] = 0.f; }
            // attraction: 2ab d^(b-1) / (1 + a d^b) = 2ab d^b / (d (1 + a d^b)), 0 where d <= 0 (umap.py:252-256): d = 0 makes d^b = 0
            // and the guarded reciprocal finite, so the coefficient is 0 without a test -- and a slot beyond the part's count reads
            // the row itself (d = 0): no masks in the loop.  Rounds of EIGHT entries (two 16-byte list reads; the list carries 64
            // entries of slack), the entries of round r + 1 requested before the gathers of round r are issued, and between issuing
            // the gathers of a round and using them the lane evaluates the NEGATIVES that belong to those entries (five per fired
            // edge, served from LDS: pure vector work): -2b / ((d + eps)(1 + a d^b)) (umap.py:272-281).  With the reference's rate
            // of 5 (and a cap that is a multiple of it) the negatives come in whole groups of five: no masks there either
            i32x4 ln[2];
            ln[0] = ln[1] = i32x4{(int)gi, (int)gi, (int)gi, (int)gi};
            if (e1 > e0 && !ab_list) { __builtin_memcpy(&ln[0], lst + e0, 16); __builtin_memcpy(&ln[1], lst + e0 + 4, 16); }
            for (int k = e0; k < e1; k += 2 * U) {
                const i32x4 lc[2] = {ln[0], ln[1]};
                if (k + 2 * U < e1 && !ab_list) { __builtin_memcpy(&ln[0], lst + k + 2 * U, 16); __builtin_memcpy(&ln[1], lst + k + 3 * U, 16); }
                Vec<NC> zj[2 * U];
#pragma unroll
                for (int u = 0; u < 2 * U; ++u) {
                    // a slot beyond the part's count takes no part in the gather (offset outside the descriptor's range: no request
                    // and no branch -- a branch per slot breaks the eight loads into eight issue / wait groups)
                    const bool valid = k + u < e1;
                    const uint32_t jc = (uint32_t)lc[u >> 2][u & 3];
                    if (DBG && ab_gather) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) zj[u].v[c] = zi.v[c] + __uint_as_float((jc & 0xffffu) | 0x3f000000u);
                    } else {
                        zj[u] = gather_z<NC>(zrs, valid ? jc * (uint32_t)(NC * 4) : 0xffffffffu);
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c) zj[u].v[c] = valid ? zj[u].v[c] : zi.v[c];
                }
                if (P.exact5) {
#pragma unroll 1
                    for (int g = 0; g < 2 * U && kn < kn1; ++g, kn += 5) {
                        Vec<NC> zn[5];
#pragma unroll
                        for (int u = 0; u < 5; ++u) { zn[u] = pool_read<NC>(pool, x >> (32 - LOGP)); x += rk.beta; }
#pragma unroll
                        for (int u = 0; u < 5; ++u) {
                            float df[NC];
                            const float d = sqdist_fma<NC>(zi, zn[u], df);
                            const float coef = m2b * fast_rcp((d + P.eps) * __builtin_fmaf(P.a, fast_pow(d, P.b), 1.0f));
#pragma unroll
                            for (int c = 0; c < NC; ++c) pr[c] = __builtin_fmaf(coef, df[c], pr[c]);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2 * U; ++u) {
                    float df[NC];
                    const float d = sqdist_fma<NC>(zi, zj[u], df);
                    const float pb = fast_pow(d, P.b);
                    const float coef = pb * two_ab * fast_rcp(fmaxf(d * __builtin_fmaf(P.a, pb, 1.0f), 1e-37f));
#pragma unroll
                    for (int c = 0; c < NC; ++c) pa[c] = __builtin_fmaf(coef, df[c], pa[c]);
                }
            }
            // what is left of the part's negatives (a rate other than 5, or more items than five per listed edge)
            for (; kn < kn1; kn += U) {
                Vec<NC> zn[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { zn[u] = pool_read<NC>(pool, x >> (32 - LOGP)); x += rk.beta; }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float df[NC];
                    const float d = sqdist_fma<NC>(zi, zn[u], df);
                    float coef = m2b * fast_rcp((d + P.eps) * __builtin_fmaf(P.a, fast_pow(d, P.b), 1.0f));
                    if (!(kn + u < kn1)) coef = 0.f;
#pragma unroll
                    for (int c = 0; c < NC; ++c) pr[c] = __builtin_fmaf(coef, df[c], pr[c]);
                }
            }
            if (slot0 >= 0) {       // a part of a split row: its sums go to the row's slot
#pragma unroll
                for (int c = 0; c < NC; ++c) { psum[(slot0 + j) * 2 * NC + c] = pa[c]; psum[(slot0 + j) * 2 * NC + NC + c] = pr[c]; }
            } else if (j == 0) {    // parts are ADDED in order (the same association as the split form)
#pragma unroll
                for (int c = 0; c < NC; ++c) { ga[c] = pa[c]; gr[c] = pr[c]; }
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) { ga[c] += pa[c]; gr[c] += pr[c]; }
            }
        }
        if (slot0 < 0) q2 += pool_store_grad<NC>(P, r, gi64, zi, iter, inspected, ga, gr);
        if (q < 2) stamp(4 + q);
    }
    // 4. split rows: their parts' sums in part order
    __syncthreads();
    for (uint32_t sidx = (uint32_t)t; sidx < scount; sidx += THREADS) {
        const uint32_t row = srow[sidx];
        const uint32_t rx = rowx[row], m = rx & 255u, b0 = (rx >> 8) - 1u;
        const int64_t r = rowbase + row - P.row0;
        if (r < 0 || r >= P.n_rows) continue;
        float ga[NC], gr[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { ga[c] = psum[b0 * 2 * NC + c]; gr[c] = psum[b0 * 2 * NC + NC + c]; }
        for (uint32_t j = 1; j < m; ++j) {
#pragma unroll
            for (int c = 0; c < NC; ++c) { ga[c] += psum[(b0 + j) * 2 * NC + c]; gr[c] += psum[(b0 + j) * 2 * NC + NC + c]; }
        }
        q2 += pool_store_grad<NC>(P, r, rowbase + row, load_z<NC>(P.Z, rowbase + row), iter, inspected, ga, gr);
    }
    if (inspected && P.norm2) {      // the same for every thread of the launch
        q2 = wave_sum(q2);
        if (lane == 0 && q2 != 0.f) atomicAdd(&P.norm2[iter / (uint32_t)P.check_interval], q2);
    }
}

// test hook: the global row of every item the gradient kernel draws for rows with nuse[r] items (-1: beyond the row's count;
// -2: a dropped draw -- the row itself or the padding of the last run)
template <int ROWS, int RUNS, int RUNLEN>
__global__ __launch_bounds__(256) void umap_pool_debug_kernel(uint64_t seed, uint32_t iter, int64_t n_total, int64_t row0, int64_t n_rows,
                                                              const int32_t* __restrict__ nuse, int width, int64_t* __restrict__ out) {
    constexpr int LOGP = ilog2(RUNS * RUNLEN), LOGR = ilog2(RUNLEN);
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t gi = row0 + r;
    const uint32_t n_runs = (uint32_t)((n_total + RUNLEN - 1) / RUNLEN);
    const uint32_t bkey = pool_block_key(seed, iter, (uint32_t)(gi / ROWS));
    const PoolRowKey rkey = pool_row_key(seed, iter, gi);
    const int n = nuse[r];
    for (int k = 0; k < width; ++k) {
        int64_t j = -1;
        if (k < n) {
            const uint32_t s = pool_item<LOGP>(rkey, (uint32_t)k);
            j = (int64_t)pool_run(bkey, s >> LOGR, n_runs) * RUNLEN + (s & (uint32_t)(RUNLEN - 1));
            if (j >= n_total || j == gi) j = -2;
        }
        out[(size_t)r * width + k] = j;
    }
}

// launches of fewer than this many global blocks (8 wavefronts each: fewer than two per SIMD) go to the split form with ONE row per
// lane -- twice the wavefronts, half the serial work in each: 30.7 -> 24.6 us per call at 123 blocks, 32.2 -> 31.3 at 245, nothing
// from 489 blocks on (profiles/r06_pool_split.jsonl).  (Splitting alone, two rows per lane in smaller workgroups, changed nothing
// at 123 blocks: every wavefront already had a SIMD to itself and ran as long as before.)
constexpr int64_t POOL_SPLIT_BELOW = 256;
template <int NC, int THREADS, int RPT, int RUNS, int RUNLEN, bool SPLITS = false>
static int launch_pool(const PoolGradParams& P0, hipStream_t st, int split = 0) {
    const bool dbg = P0.ablate != 0 || P0.dbg_times != nullptr;
    constexpr int ROWS = THREADS * RPT;
    PoolGradParams P = P0;
    P.gb0 = P.row0 / ROWS;
    P.n_runs = (uint32_t)((P.n_total + RUNLEN - 1) / RUNLEN);
    const int64_t gb1 = (P.row0 + P.n_rows - 1) / ROWS;
    const int64_t nb = gb1 - P.gb0 + 1;
    if (dbg) hipLaunchKernelGGL((umap_pool_grad_kernel<NC, THREADS, RPT, RUNS, RUNLEN, 1, true>), dim3((unsigned)nb), dim3(THREADS), 0, st, P);
    else if (SPLITS && RPT == 2 && split == 4)
        hipLaunchKernelGGL((umap_pool_grad_kernel<NC, SPLITS ? THREADS / 2 : THREADS, SPLITS ? 1 : RPT, RUNS, RUNLEN, SPLITS ? 4 : 1, false>), dim3((unsigned)(nb * 4)), dim3(THREADS / 2), 0, st, P);
    else if (SPLITS && RPT == 2 && split == 8)
        hipLaunchKernelGGL((umap_pool_grad_kernel<NC, SPLITS ? THREADS / 4 : THREADS, SPLITS ? 1 : RPT, RUNS, RUNLEN, SPLITS ? 8 : 1, false>), dim3((unsigned)(nb * 8)), dim3(THREADS / 4), 0, st, P);
    else if (SPLITS && RPT == 2 && (split == 2 || (split == 0 && nb < POOL_SPLIT_BELOW)))
        hipLaunchKernelGGL((umap_pool_grad_kernel<NC, THREADS, SPLITS ? 1 : RPT, RUNS, RUNLEN, SPLITS ? 2 : 1, false>), dim3((unsigned)(nb * 2)), dim3(THREADS), 0, st, P);
    else hipLaunchKernelGGL((umap_pool_grad_kernel<NC, THREADS, RPT, RUNS, RUNLEN, 1, false>), dim3((unsigned)nb), dim3(THREADS), 0, st, P);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}

// geometries: threads per block, rows per thread, pool runs, rows per run -- TDR_POOL_GEOMS lists them for the debug kernel too
#define TDR_POOL_GEOMS(X) X(1, 512, 1, 256, 16) X(2, 512, 2, 256, 16) X(3, 512, 2, 128, 16) X(4, 256, 2, 256, 8) X(5, 512, 2, 512, 8) X(6, 1024, 1, 256, 16)
template <int NC>
static int launch_pool_geom(const PoolGradParams& P, int geom, hipStream_t st) {
    switch (geom) {
#define TDR_POOL_CASE(G, T, R, Q, L) case G: return launch_pool<NC, T, R, Q, L>(P, st);
        TDR_POOL_GEOMS(TDR_POOL_CASE)
#undef TDR_POOL_CASE
        // geometries 16 + s (s = 1, 2, 4): the default geometry with s workgroups per block whatever the launch size (tests, measurement)
        case 17: return launch_pool<NC, TDR_POOL_THREADS, TDR_POOL_RPT, TDR_POOL_RUNS, TDR_POOL_RUNLEN, true>(P, st, 1);
        case 18: return launch_pool<NC, TDR_POOL_THREADS, TDR_POOL_RPT, TDR_POOL_RUNS, TDR_POOL_RUNLEN, true>(P, st, 2);
        case 20: return launch_pool<NC, TDR_POOL_THREADS, TDR_POOL_RPT, TDR_POOL_RUNS, TDR_POOL_RUNLEN, true>(P, st, 4);
        case 24: return launch_pool<NC, TDR_POOL_THREADS, TDR_POOL_RPT, TDR_POOL_RUNS, TDR_POOL_RUNLEN, true>(P, st, 8);
        default: return launch_pool<NC, TDR_POOL_THREADS, TDR_POOL_RPT, TDR_POOL_RUNS, TDR_POOL_RUNLEN, true>(P, st);
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
    if (t_local < 0 || t_local >= 32 || neg_rate < 0 || n_negatives < 0 || !tdr_pool_geom_ok(geom)) return TDR_ERR_BAD_ARG;
    if (((uintptr_t)Z & 15u) != 0 || n_total * nc * 4 >= 0xffffffffLL) return TDR_ERR_BAD_ARG;   // Z behind one buffer descriptor
    if (!tdr_umap_pool_supported(nc)) return TDR_ERR_UNSUPPORTED;
    PoolGradParams P = {};
    P.Z = Z; P.nc = nc; P.n_total = n_total; P.row0 = row0; P.n_rows = n_rows; P.list = list; P.hdr = (const uint2*)hdr;
    P.t_local = t_local; P.a = a; P.b = b; P.neg_rate = neg_rate; P.n_negatives = n_negatives; P.seed = seed; P.iter = (uint32_t)n_iter;
    P.iter_base = nullptr; P.exag = exag; P.rep = rep; P.eps = eps; P.grad = grad;
    P.exact5 = (neg_rate == 5 && n_negatives % 5 == 0) ? 1 : 0;
    return launch_pool_grad(P, geom, (hipStream_t)stream);
}

/* tdr_umap_pool_grad_f32 with torch.optim.SGD's step in the same launch (affinity_matcher.py:427, momentum 0): the stepped rows
 * z - lr g go to Z_out (the OTHER embedding buffer, same shape as Z: every row of the launch reads the old positions), nan_flag
 * as tdr_sgd_step_f32 sets it; grad may be NULL (nobody reads the gradient of this iteration). */
int tdr_umap_pool_grad_step_f32(const float* Z, float* Z_out, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* list,
                                const void* hdr, int t_local, float a, float b, int n_iter, int neg_rate, int n_negatives, uint64_t seed,
                                float exag, float rep, float eps, float* grad, float lr, int* nan_flag, int geom, void* stream) {
    if (!Z || !Z_out || Z == Z_out || !list || !hdr || !nan_flag || n_rows <= 0 || row0 < 0 || n_total < 2 || n_total >= 0x7fffffffLL) return TDR_ERR_BAD_ARG;
    if (t_local < 0 || t_local >= 32 || neg_rate < 0 || n_negatives < 0 || !tdr_pool_geom_ok(geom)) return TDR_ERR_BAD_ARG;
    if (((uintptr_t)Z & 15u) != 0 || ((uintptr_t)Z_out & 7u) != 0 || n_total * nc * 4 >= 0xffffffffLL) return TDR_ERR_BAD_ARG;
    if (!tdr_umap_pool_supported(nc)) return TDR_ERR_UNSUPPORTED;
    PoolGradParams P = {};
    P.Z = Z; P.nc = nc; P.n_total = n_total; P.row0 = row0; P.n_rows = n_rows; P.list = list; P.hdr = (const uint2*)hdr;
    P.t_local = t_local; P.a = a; P.b = b; P.neg_rate = neg_rate; P.n_negatives = n_negatives; P.seed = seed; P.iter = (uint32_t)n_iter;
    P.iter_base = nullptr; P.exag = exag; P.rep = rep; P.eps = eps; P.grad = grad; P.Z_out = Z_out; P.lr = lr; P.nan_flag = nan_flag;
    P.exact5 = (neg_rate == 5 && n_negatives % 5 == 0) ? 1 : 0;
    return launch_pool_grad(P, geom, (hipStream_t)stream);
}

/* measurement hook: one launch of the DBG instance with the switches `ablate` (1 no pool staging, 2 no attraction, 4 no negatives,
 * 8 rows not sorted, 16 no gathers, 32 no list reads); times (optional): 8 uint64 shader-clock stamps per wavefront (block-major) -- start, before / after the first barrier, after the sort, after each row pass. */
int tdr_umap_pool_grad_debug_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* list, const void* hdr,
                                 int t_local, float a, float b, int n_iter, int neg_rate, int n_negatives, uint64_t seed, float* grad,
                                 int geom, int ablate, void* times, void* stream) {
    if (!Z || !list || !hdr || !grad || n_rows <= 0 || row0 < 0 || n_total < 2 || n_total >= 0x7fffffffLL) return TDR_ERR_BAD_ARG;
    if (t_local < 0 || t_local >= 32 || !tdr_pool_geom_ok(geom) || ((uintptr_t)Z & 15u) != 0) return TDR_ERR_BAD_ARG;
    if (!tdr_umap_pool_supported(nc)) return TDR_ERR_UNSUPPORTED;
    PoolGradParams P = {};
    P.Z = Z; P.nc = nc; P.n_total = n_total; P.row0 = row0; P.n_rows = n_rows; P.list = list; P.hdr = (const uint2*)hdr;
    P.t_local = t_local; P.a = a; P.b = b; P.neg_rate = neg_rate; P.n_negatives = n_negatives; P.seed = seed; P.iter = (uint32_t)n_iter;
    P.exag = 1.f; P.rep = 1.f; P.eps = 1e-3f; P.grad = grad; P.ablate = ablate | 0x40000000; P.dbg_times = (unsigned long long*)times;
    P.exact5 = (neg_rate == 5 && n_negatives % 5 == 0) ? 1 : 0;
    return launch_pool_grad(P, geom, (hipStream_t)stream);
}

int tdr_umap_pool_debug_negatives(uint64_t seed, int n_iter, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nuse, int geom,
                                  int width, int64_t* out, void* stream) {
    if (!nuse || !out || n_rows <= 0 || width <= 0 || n_total < 2 || !tdr_pool_geom_ok(geom)) return TDR_ERR_BAD_ARG;
    const dim3 grid((unsigned)((n_rows + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
#define TDR_POOL_DBG(R, Q, L) hipLaunchKernelGGL((umap_pool_debug_kernel<R, Q, L>), grid, dim3(256), 0, st, seed, (uint32_t)n_iter, n_total, row0, n_rows, nuse, width, out)
    switch (geom) {
#define TDR_POOL_CASE(G, T, R, Q, L) case G: TDR_POOL_DBG(T * R, Q, L); break;
        TDR_POOL_GEOMS(TDR_POOL_CASE)
#undef TDR_POOL_CASE
        default: TDR_POOL_DBG(TDR_POOL_THREADS * TDR_POOL_RPT, TDR_POOL_RUNS, TDR_POOL_RUNLEN); break;
    }
#undef TDR_POOL_DBG
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
