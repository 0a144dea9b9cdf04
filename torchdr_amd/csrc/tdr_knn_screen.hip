// K1s -- two-stage exact kNN for gfx950: fp16-split SCREENING on the 2.5 PFLOP/s matrix pipe, then
// exact fp32 RESCORING of the few survivors.  Results are bit-identical to the one-stage exact kernel
// (tdr_knn.hip) and therefore to the reference CPU backend (distance/torch.py:82-120 + utils/utils.py:215).
//
// Why it is exact.  Let d_j be the distance the reference computes in fp32 and a_j the screening value
//     a_j = (||x||^2 + ||y_j||^2) - 2 * (h.h' + h.l' + l.h'),        x = h + l + eps,  h, l in fp16
// (h = fp16(s x), l = fp16(s x - h), s a power of two chosen so that max |s x| lies in [2^13, 2^14): the split
// keeps 22 significant bits and the scaling is exact).  For every pair |a_j - d_j| <= E_q with
//     E_q = c_rel * ||x_q|| * max_j ||y_j|| + c_abs * (||x_q||^2 + max_j ||y_j||^2) + c_den
// (screen_band below: fp16-split representation error 3 * 2^-22, fp32 accumulation of 3*D products in ANY
// order with a possibly truncating adder, the reference's own fp32 rounding -- all as worst-case bounds).  If a_(k) is the k-th smallest
// screening value, the true k-th smallest distance is <= a_(k) + E_q, so every true neighbour has
// a_j <= a_(k) + 2 E_q: the candidate set {j : a_j <= a_(k) + 2 E_q} contains the exact top-k.  The scan keeps
// the L >= k smallest screening values per query (L - k spare slots); if the spare slots overflow (list full
// and a_(L) <= a_(k) + 2 E_q) the query is flagged and the caller re-runs it through the one-stage exact
// kernel.  Survivors are re-evaluated by knn_rescore_kernel with the reference's arithmetic (k-ordered fmaf
// chain, same association of the norm sum) and ranked by the canonical (distance, index) key.
//
// Cost: 3 v_mfma_f32_32x32x16_f16 per 16 features (24 x 32 cycles per 32x32 tile at D = 128) instead of
// 64 x 64 cycles of v_mfma_f32_32x32x2_f32 -- 5.3x fewer matrix-pipe cycles -- with the same 16 KiB tile image,
// LDS-DMA staging and software-pipelined epilogue as the one-stage kernel.
#include "tdr_common.h"
#include "tdr_knn_screen_common.h"
#include <stdlib.h>

namespace tdr {
namespace scr {

#ifdef TDR_SCREEN_STATS
// measurement build only (tools/screen_stats.py): [0] wave cycles of knn_screen_kernel, [1] cycles inside the list updates,
// [2] merge events (query, tile), [3] survivors merged, [4] serial insertions tried, [5] tile steps, [6] cycles at workgroup barriers
__device__ unsigned long long g_screen_stats[8][64];   // 64 slots per counter (by workgroup): the host adds them up
#define TDR_STAT_ADD(i, v) do { if (C.lane == 0) atomicAdd(&g_screen_stats[i][blockIdx.x & 63], (unsigned long long)(v)); } while (0)
#else
#define TDR_STAT_ADD(i, v) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------
// meta reductions
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// strided variant for row-padded inputs (ldx > d)
__global__ __launch_bounds__(256) void absmax2d_kernel(const float* __restrict__ x, int64_t n, int d, int64_t ldx,
                                                       uint32_t* __restrict__ out) {
    uint32_t m = 0;
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        m = max(m, __float_as_uint(x[r * ldx + c]) & 0x7fffffffu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// ---------------------------------------------------------------------------------------------------------
// pack16_kernel: X (n x d fp32) -> fp16-split tile images.  Block (2*s + term) of tile b is 1 KiB:
//   lane (g*32 + i), element e  <-  term(s_scale * X[32b + i][16 s + 8 g + e]),   term 0 = h, term 1 = l
// i.e. exactly the A/B fragment of v_mfma_f32_32x32x16_f16 for K-slice s (both operands use the same
// lane -> k map, so any consistent element order yields the dot product).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack16_kernel(const float* __restrict__ X, int64_t n, int d, int64_t ldx, int ks,
                                                     const float* __restrict__ norms, const uint32_t* __restrict__ meta,
                                                     const int32_t* __restrict__ row_map, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int dimg = ks * 16;
    const int ld = dimg + 4;
    const int64_t row0 = (int64_t)blockIdx.x * TILE_ROWS;
    const int tid = threadIdx.x;
    const float s = pow2f(scale_exp(meta[0]));
    // image row r' holds source row row_map[r'] (-1 = padding row) when a map is given (cluster-sorted order)
    for (int idx = tid; idx < TILE_ROWS * dimg; idx += 256) {
        const int r = idx / dimg, c = idx - r * dimg;
        float v = 0.f;
        if (row0 + r < n && c < d) {
            const int64_t src = row_map ? (int64_t)row_map[row0 + r] : row0 + r;
            if (src >= 0) v = X[(size_t)src * ldx + c];
        }
        xs[r * ld + c] = v * s;  // exact (power of two; inputs are finite and far from the fp32 range ends)
    }
    __syncthreads();
    char* img = reinterpret_cast<char*>(out + (size_t)blockIdx.x * tile16_stride_floats(ks));
    for (int idx = tid; idx < ks * 64; idx += 256) {
        const int sl = idx >> 6, l = idx & 63, g = l >> 5, i = l & 31;
        const float* xr = xs + i * ld + 16 * sl + 8 * g;
        f16x8 hv, lv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = xr[e];
            const _Float16 h = (_Float16)v;              // round to nearest even
            const float r = __fsub_rn(v, (float)h);       // exact
            hv[e] = h;
            lv[e] = (_Float16)r;
        }
        *reinterpret_cast<f16x8*>(img + (size_t)(2 * sl) * 1024 + l * 16) = hv;
        *reinterpret_cast<f16x8*>(img + (size_t)(2 * sl + 1) * 1024 + l * 16) = lv;
    }
    if (tid < 64) {
        float* nb = reinterpret_cast<float*>(img + (size_t)ks * 2048);
        float v = 0.f;
        if (tid < 32) {
            v = __builtin_inff();
            if (row0 + tid < n) {
                const int64_t src = row_map ? (int64_t)row_map[row0 + tid] : row0 + tid;
                if (src >= 0) v = norms[src];
            }
        }
        nb[tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Screening scan
// ---------------------------------------------------------------------------------------------------------
struct ScreenParams {
    const float* qp;      // fp16-split query images
    const float* yp;      // fp16-split database images
    const uint32_t* meta;
    int64_t nq, q_offset, n_db;
    int k;                // neighbours wanted
    int L;                // list length (k <= L)
    int exclude_self;
    int n_db_tiles, tiles_per_split, n_splits;
    int dpad;
    int terms;            // 3: h.h' + h.l' + l.h'; 1: h.h' only (wider band, a third of the matrix work)
    uint64_t* cand;       // (n_splits, nq, L) ascending screening keys
    // cluster-bound pruning (self search on cluster-sorted, tile-padded points; all NULL = visit every tile)
    int batch0;           // first query batch of this launch (a rank searches its range of the sorted order)
    int n_clusters;
    const int32_t* tile_cluster;    // (n_db_tiles) cluster of every 32-row tile (queries and database share the order)
    const int32_t* clus_tile_begin; // (n_clusters + 1) first tile of each cluster
    const float* clus_radius;       // (n_clusters) max member distance to the centre, rounded up
    const float* clus_dist;         // (n_clusters, n_clusters) centre distances, rounded down
    const int32_t* clus_order;      // (n_clusters, n_clusters) clusters by increasing centre distance (self first)
    int max_visit;                  // approximate (IVF-style) search: clusters a workgroup may scan at most; 0 = exact
    const float* tile_cdist;        // optional (n_db_tiles, n_clusters): lower bound of min over the tile's rows of |x - c_c| --
                                    // the bound |x - y| >= |x - c_c| - R_c of the tile's own rows replaces the ball-to-ball bound
    int32_t* lost;                  // LAZY kernels: (nq, caller-zeroed) 1 = the query's error band held more candidates than its buffer
    int L_out;                      // entries written per (split, query) list (row stride of `cand`); LAZY kernels with L_out < L compact
                                    // every buffer once more at the end and report a query whose band population exceeds L_out as lost
};

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// Cooperative sorted insertion into an ascending len-entry list (lane p owns entries p, p + 64).  Returns the
// new k-th (position kpos) and last (position len - 1) keys when the candidate entered.
template <int ITEMS>
__device__ __forceinline__ bool coop_insert2(uint64_t* Lst, int len, int kpos, uint64_t cand, int lane, uint64_t& new_kth,
                                             uint64_t& new_tail) {
    uint64_t cur[ITEMS], prev[ITEMS], nv[ITEMS];
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        const int p = lane + 64 * t;
        cur[t] = (p < len) ? Lst[p] : KEY_SENTINEL;
        prev[t] = (p > 0 && p < len) ? Lst[p - 1] : 0ull;
    }
    const int tl = (len - 1) & 63, ti = (len - 1) >> 6;
    const int kl = kpos & 63, ki = kpos >> 6;
    uint64_t tk = 0;
#pragma unroll
    for (int t = 0; t < ITEMS; ++t)
        if (t == ti) tk = readlane_u64(cur[t], tl);
    if (cand >= tk) return false;  // wave-uniform
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        const int p = lane + 64 * t;
        nv[t] = (cur[t] < cand) ? cur[t] : ((p == 0 || prev[t] < cand) ? cand : prev[t]);
        if (p < len && nv[t] != cur[t]) Lst[p] = nv[t];
    }
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        if (t == ti) new_tail = readlane_u64(nv[t], tl);
        if (t == ki) new_kth = readlane_u64(nv[t], kl);
    }
    return true;
}

template <int QB>
struct SCtx {
    const ScreenParams* P;
    uint64_t* keys;  // this wave's lists [QB][32][L]
    int lane, q, h;
    int64_t qt0;     // first query tile of this wave (tiles qt0 .. qt0 + QB - 1)
    float xn[QB], band[QB], m2s;
};

// The hot filter works on the REDUCED value c' = ||y||^2 - 2 s^-2 acc (one fma per candidate) against the
// lane's reduced threshold tau_r = (tau - ||x||^2) rounded up; only a survivor gets its full screening value
// a = c' + ||x||^2, which is what the lists hold.
__device__ __forceinline__ float reduce_tau(float tau, float xn) {
    return (tau - xn) + 2.3841858e-07f * (fabsf(tau) + xn);  // + 4u (|tau| + xn): never rejects an a <= tau
}

// one candidate at a time: the cheaper form while survivors are rare (steady state of a long scan)
template <int ITEMS, int QB>
__device__ __forceinline__ void screen_insert_serial(const SCtx<QB>& C, const float (&dv)[QB][16], const float (&pmin)[QB][4],
                                              int Tprev, float (&tau_r)[QB]) {
    const ScreenParams& P = *C.P;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (!__any(pmin[qb][g] <= tau_r[qb])) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                unsigned long long m = __ballot(dv[qb][r] <= tau_r[qb]);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const int sq = src & 31;
                    const int64_t j = (int64_t)Tprev * 32 + 4 * (src >> 5) + e + 8 * g;
                    if (j >= P.n_db || (P.exclude_self && j == (C.qt0 + qb) * 32 + sq + P.q_offset)) continue;
                    const float cred =
                        __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dv[qb][r]), src));
                    if (!(cred < __builtin_inff())) continue;  // padding row (norm +inf)
                    const float xq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, C.xn[qb]), sq));
                    uint64_t nk, nt;
                    TDR_STAT_ADD(4, 1);
                    if (coop_insert2<ITEMS>(C.keys + ((size_t)qb * 32 + sq) * P.L, P.L, P.k - 1, mkkey(cred + xq, (uint32_t)j),
                                            C.lane, nk, nt)) {
                        if (C.q == sq)
                            tau_r[qb] = reduce_tau(fminf(u2f((uint32_t)(nk >> 32)) + C.band[qb], u2f((uint32_t)(nt >> 32))),
                                                   C.xn[qb]);
                    }
                }
            }
        }
    }
}

// Survivors of one finished tile -> the queries' lists.  A lane holds 16 candidate values of ONE query (two lanes per
// query: the tile's rows 4h + (r & 3) + 8 (r >> 2)).  Per query with at least one survivor the wavefront MERGES all of
// that query's survivors (<= 32) into its ascending list in one pass:
//   collect  the survivors' keys are pulled out of the owning lanes' registers through the scalar unit
//            (v_readlane, select on the lane id) into lanes 0 .. nb-1;
//   rank     every list entry counts the survivors below it (its shift), every survivor counts the list entries
//            (ballot + popcount) and the other survivors below it (its position) -- one u64 compare per (entry,
//            survivor) pair and no dependent shuffle chains;
//   place    entries and survivors are scattered to their final positions (LDS), entries pushed past the end fall off.
// The result is the L smallest of (list + survivors), exactly what inserting them one by one produces, at
// ~(300 + 56 nb) cycles per query instead of ~450 per survivor: on clustered data a query meets most of its candidates
// in the first tiles of its own cluster, where nearly every row still beats the threshold.
template <int ITEMS, int QB>
__device__ __forceinline__ void screen_insert_merge(const SCtx<QB>& C, const float (&dv)[QB][16], const float (&pmin)[QB][4],
                                              int Tprev, float (&tau_r)[QB]) {
    const ScreenParams& P = *C.P;
    const int L = P.L;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int64_t jb = (int64_t)Tprev * 32 + 4 * C.h;
        const int64_t jself = (C.qt0 + qb) * 32 + C.q + P.q_offset;
        // opaque copies: nothing below may be speculated above the (rarely taken) branch that leads here
        float tq = tau_r[qb], xq = C.xn[qb];
        asm volatile("" : "+v"(tq), "+v"(xq));
        uint32_t hi[16];
        uint32_t smask = 0;  // bit r: this lane's candidate r survives (one VGPR instead of 16 ballots in SGPRs)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t j = jb + (r & 3) + 8 * (r >> 2);
            // padding rows carry +inf norms; rows beyond the database and the query's own row are not candidates
            const bool ok = dv[qb][r] <= tq && dv[qb][r] < __builtin_inff() && j < P.n_db && !(P.exclude_self && j == jself);
            smask |= ok ? (1u << r) : 0u;
            hi[r] = f2u(dv[qb][r] + xq);  // the full screening value a = c' + ||x||^2 is what the lists hold
        }
        const unsigned long long any = __ballot(smask != 0u);
        uint32_t qmask = (uint32_t)any | (uint32_t)(any >> 32);
        while (qmask) {
            const int sq = __builtin_amdgcn_readfirstlane(__builtin_ctz(qmask));
            qmask &= qmask - 1u;
            uint64_t* Lst = C.keys + ((size_t)qb * 32 + sq) * L;
            uint64_t cur[ITEMS];
#pragma unroll
            for (int t = 0; t < ITEMS; ++t) {
                const int p = C.lane + 64 * t;
                cur[t] = (p < L) ? Lst[p] : KEY_SENTINEL;
            }
            // collect
            uint32_t Bhi = (uint32_t)(KEY_SENTINEL >> 32), Blo = 0xffffffffu;
            int nb = 0;
            const uint32_t m2[2] = {(uint32_t)__builtin_amdgcn_readlane((int)smask, sq),
                                    (uint32_t)__builtin_amdgcn_readlane((int)smask, sq + 32)};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    if ((m2[hh] >> r) & 1u) {
                        const uint32_t kh = (uint32_t)__builtin_amdgcn_readlane((int)hi[r], sq + 32 * hh);
                        const uint32_t kl = (uint32_t)(Tprev * 32 + 4 * hh + (r & 3) + 8 * (r >> 2));
                        const bool mine = C.lane == nb;  // lane nb receives survivor nb
                        Bhi = mine ? kh : Bhi;
                        Blo = mine ? kl : Blo;
                        ++nb;
                    }
                }
            }
            TDR_STAT_ADD(2, 1);
            TDR_STAT_ADD(3, nb);
            // rank
            const uint64_t Bv = ((uint64_t)Bhi << 32) | (uint64_t)Blo;
            int cntS = 0, rankB = 0;
            int shift[ITEMS];
#pragma unroll
            for (int t = 0; t < ITEMS; ++t) shift[t] = 0;
            for (int t2 = 0; t2 < nb; ++t2) {
                const uint64_t bk = readlane_u64(Bv, t2);
                int c = 0;
#pragma unroll
                for (int t = 0; t < ITEMS; ++t) {
                    const bool lt = cur[t] < bk;
                    c += __popcll(__ballot(lt));
                    shift[t] += lt ? 0 : 1;
                }
                cntS = (C.lane == t2) ? c : cntS;
                rankB += (bk < Bv) ? 1 : 0;
            }
            // place
#pragma unroll
            for (int t = 0; t < ITEMS; ++t) {
                const int p = C.lane + 64 * t;
                const int np = p + shift[t];
                if (p < L && np < L && shift[t] > 0) Lst[np] = cur[t];
            }
            if (C.lane < nb) {
                const int pos = rankB + cntS;
                if (pos < L) Lst[pos] = Bv;
            }
            // the query's new threshold (LDS operations of a wavefront complete in order: these reads see the writes)
            const uint64_t nk = Lst[P.k - 1], nt = Lst[L - 1];
            if (C.q == sq)
                tau_r[qb] = reduce_tau(fminf(u2f((uint32_t)(nk >> 32)) + C.band[qb], u2f((uint32_t)(nt >> 32))), C.xn[qb]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// LAZY candidate buffers (round 6, the cluster-pruned scan).  The sorted lists above pay per (query, tile) with a survivor:
// on the headline data a query meets survivors in 27 of the 32 tiles of its own cluster (5.7 each: neighbours and
// non-neighbours of one blob are a few per cent apart), ~300 + 56 nb cycles every time, one query after the other -- 61 % of
// the wave cycles of the pruned scan (tools/screen_stats.py).  Here a query's region is an UNSORTED buffer of L entries
// (L as large as the LDS of a whole CU allows: 126 at D <= 128) with its count in a register of the query's two lanes:
//   append   every lane stores its own survivors at (count + its offset inside the lane pair) -- all 32 queries of the
//            wavefront at once, no ranking, no shuffling;
//   compact  only when a buffer cannot take a tile's survivors (and at the end of a cluster, to refresh the threshold the
//            pruning test reads): an upper bound U of the k-th smallest screening value of the buffer by bisection on the
//            ordered bit images (ballot + popcount per step), tau = U + 2E, entries above tau dropped, the rest moved up.
// Exactness is the list form's argument: tau_q >= a_(k)(everything seen) + 2E at every moment (U >= a_(k) of a subset of
// what was seen), so no true neighbour is ever refused or dropped; a buffer that is still full after a compaction holds L
// candidates inside the band -- the query is marked lost and the host recomputes it (the sorted lists flag the same state).
// ---------------------------------------------------------------------------------------------------------
struct LazyState {
    int cnt;     // entries in the buffer of this lane's query (the same in both lanes of the query)
    int fresh;   // entries appended since the last compaction
    int lost;
};

// lane primitives of the lazy buffers (tools/lab/lane_test.hip checks them on the device)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// value of the other lane of a query's pair (lane ^ 32): v_permlane32_swap instead of a trip through the LDS crossbar
__device__ __forceinline__ int pair_other(int v, int h) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(h ? r[0] : r[1]);
}
// wave-wide max / min of unsigned values by DPP row shifts and row broadcasts (the result lands in lane 63)
template <bool MAX>
__device__ __forceinline__ uint32_t wave_extreme_u32(uint32_t v) {
#define TDR_DPP_STEP(CTRL, RM)                                                                                     \
    {                                                                                                              \
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, RM, 0xf, false);            \
        v = MAX ? (o > v ? o : v) : (o < v ? o : v);                                                               \
    }
    TDR_DPP_STEP(0x111, 0xf) TDR_DPP_STEP(0x112, 0xf) TDR_DPP_STEP(0x114, 0xf) TDR_DPP_STEP(0x118, 0xf)
    TDR_DPP_STEP(0x142, 0xa) TDR_DPP_STEP(0x143, 0xc)
#undef TDR_DPP_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// compaction of the buffer of query sq (n entries, k <= n, wave-uniform); returns the new count
template <int ITEMS>
__device__ __forceinline__ int lazy_compact(const SCtx<1>& C, int sq, int n, float& tau_new, bool exact = false) {
    const ScreenParams& P = *C.P;
#ifdef TDR_SCREEN_STATS
    const unsigned long long tc0 = __builtin_readcyclecounter();
    int n_it = 0;
#endif
    uint64_t* Lst = C.keys + (size_t)sq * P.L;
    uint64_t cur[ITEMS];
    uint32_t v[ITEMS];      // bit images; 0xffffffff (above every image) where the lane holds no entry
    uint32_t vmin = 0xffffffffu, vmax = 0u;
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        const int p = C.lane + 64 * t;
        const bool valid = p < n;
        cur[t] = valid ? Lst[p] : ~0ull;
        v[t] = (uint32_t)(cur[t] >> 32);
        vmin = v[t] < vmin ? v[t] : vmin;
        if (valid) vmax = v[t] > vmax ? v[t] : vmax;
    }
    uint32_t lo = wave_extreme_u32<false>(vmin), hi = wave_extreme_u32<true>(vmax);
    // invariant: at least k entries have an image <= hi.  The bisection ends as soon as a bound holds k .. k + slack entries (a few
    // entries more than the k smallest stay in the buffer: nothing next to the band's own population) -- about log2(n / slack)
    // steps on values spread over the range (4.1 on the headline data), each a compare + ballot + popcount per 64 entries.  A
    // wavefront that has its SIMD to itself issues an instruction every ~5 cycles, so what counts is their number: the first
    // form ran 20 steps behind two bpermute reductions and cost ~3.5 k cycles per compaction (tools/screen_stats.py), the early
    // end brought 2.35 k, DPP reductions the rest.
    const int slack = exact ? 0 : (P.k >= 32 ? P.k >> 3 : 4);    // exact: the bisection runs on to the k-th smallest image itself
    for (int it = 0; it < 32 && lo < hi; ++it) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        int c = 0;
#ifdef TDR_SCREEN_STATS
        ++n_it;
#endif
#pragma unroll
        for (int t = 0; t < ITEMS; ++t) c += __popcll(ballot64(v[t] <= mid));
        if (c >= P.k) {
            hi = mid;
            if (c <= P.k + slack) break;
        } else {
            lo = mid + 1u;
        }
    }
    const float band = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, C.band[0]), sq));
    tau_new = u2f(hi) + band;
    int base = 0;
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        const bool keep = v[t] != 0xffffffffu && u2f(v[t]) <= tau_new;
        const unsigned long long mk = ballot64(keep);
        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0));
        if (keep) Lst[pos] = cur[t];     // pos <= own position, and every entry was read above (LDS operations of a wavefront are in order)
        base += __popcll(mk);
    }
    TDR_STAT_ADD(2, 1);
#ifdef TDR_SCREEN_STATS
    TDR_STAT_ADD(3, __builtin_readcyclecounter() - tc0);
    TDR_STAT_ADD(4, n_it);
#endif
    return base;
}

// the finished tile's survivors into the buffers
template <int ITEMS>
__device__ __forceinline__ void screen_append_lazy(const SCtx<1>& C, const float (&dv)[1][16], int Tprev, float (&tau_r)[1], LazyState& S) {
    const ScreenParams& P = *C.P;
    const int L = P.L;
    const uint32_t jb = (uint32_t)Tprev * 32u + 4u * (uint32_t)C.h;
    const float xq = C.xn[0];
    // candidates that can never enter: padding rows carry +inf norms (their reduced value is +inf: refused by a finite threshold,
    // and the threshold is clamped to the largest finite value); rows beyond the database and the query's own row are rare and
    // wave-uniform to detect (a tile at the end of the database, the query tile's own tile)
    float tq = fminf(tau_r[0], 3.4028234663852886e38f);
    uint32_t dead = 0u;
    const int64_t self0 = C.qt0 * 32 + P.q_offset;      // the wavefront's own rows: [self0, self0 + 32)
    if ((int64_t)Tprev * 32 + 32 > P.n_db || (P.exclude_self && (int64_t)Tprev * 32 + 31 >= self0 && (int64_t)Tprev * 32 <= self0 + 31)) {
        const int64_t jself = self0 + C.q;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t j = (int64_t)jb + (r & 3) + 8 * (r >> 2);
            dead |= (j >= P.n_db || (P.exclude_self && j == jself)) ? (1u << r) : 0u;
        }
    }
    uint32_t smask = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) smask |= (dv[0][r] <= tq) ? (1u << r) : 0u;
    smask &= ~dead;
    uint64_t* Lst = C.keys + (size_t)C.q * L;
    for (;;) {
        if (ballot64(smask != 0u) == 0ull) break;
        const int n_mine = __popc(smask);
        const int n_oth = pair_other(n_mine, C.h);
        int pos = S.cnt + (C.h ? n_oth : 0);    // the lower lane of the pair stores first
        if (ballot64(S.cnt + n_mine + n_oth > L) == 0ull) {
            // every buffer of the wavefront takes its survivors (all but the few steps that fill one): no bound checks
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((smask >> r) & 1u) {
                    Lst[pos] = mkkey(dv[0][r] + xq, jb + (uint32_t)((r & 3) + 8 * (r >> 2)));
                    ++pos;
                }
            }
            S.cnt += n_mine + n_oth;
            S.fresh += n_mine + n_oth;
            break;
        }
        int wrote = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (((smask >> r) & 1u) && pos < L) {
                Lst[pos] = mkkey(dv[0][r] + xq, jb + (uint32_t)((r & 3) + 8 * (r >> 2)));
                ++pos;
                ++wrote;
                smask &= ~(1u << r);
            }
        }
        wrote += pair_other(wrote, C.h);
        S.cnt += wrote;
        S.fresh += wrote;
        const unsigned long long pend = ballot64(smask != 0u);
        if (pend == 0ull) break;
        // a query with survivors left over has a full buffer: compact it, then test the leftovers against the new threshold
        uint32_t qmask = (uint32_t)pend | (uint32_t)(pend >> 32);
        while (qmask) {
            const int sq = __builtin_amdgcn_readfirstlane(__builtin_ctz(qmask));
            qmask &= qmask - 1u;
            float tau_new;
            const int newn = lazy_compact<ITEMS>(C, sq, L, tau_new);
            if (C.q == sq) {
                if (newn >= L) {
                    // L candidates inside the band: nothing can be dropped and nothing more can be taken
                    S.lost = 1;
                    smask = 0u;
                    tau_r[0] = -__builtin_inff();
                } else {
                    S.cnt = newn;
                    S.fresh = 0;
                    tau_r[0] = reduce_tau(tau_new, xq);
                }
            }
        }
        tq = fminf(tau_r[0], 3.4028234663852886e38f);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (((smask >> r) & 1u) && !(dv[0][r] <= tq)) smask &= ~(1u << r);
    }
}

// thresholds brought up to date (end of a cluster): the queries that took at least min_fresh entries since their last compaction,
// or that have met k candidates and still have no threshold
template <int ITEMS>
__device__ __forceinline__ void lazy_refresh(const SCtx<1>& C, float (&tau_r)[1], LazyState& S, int min_fresh, bool exact = false) {
    const ScreenParams& P = *C.P;
    const bool want = !S.lost && S.cnt >= P.k && (exact || S.fresh >= min_fresh || (S.fresh > 0 && tau_r[0] == __builtin_inff()));
    const unsigned long long m = __ballot(want);
    uint32_t qmask = (uint32_t)m | (uint32_t)(m >> 32);
    while (qmask) {
        const int sq = __builtin_amdgcn_readfirstlane(__builtin_ctz(qmask));
        qmask &= qmask - 1u;
        const int n = __builtin_amdgcn_readlane(S.cnt, sq);
        float tau_new;
        const int newn = lazy_compact<ITEMS>(C, sq, n, tau_new, exact);
        if (C.q == sq) {
            S.cnt = newn;
            S.fresh = 0;
            tau_r[0] = reduce_tau(tau_new, C.xn[0]);
        }
    }
}

// few lanes with a survivor in the wavefront's tile (the steady state of a long scan): one candidate at a time
// (~450 cycles each); many (a query's first tiles, clustered data): merged per query (~800 cycles of preparation per
// tile + ~(300 + 56 nb) per query).  `lanes` = number of lanes that hold at least one survivor.
constexpr int MERGE_MIN_LANES = 8;
template <int ITEMS, int QB>
__device__ __forceinline__ void screen_insert(const SCtx<QB>& C, const float (&dv)[QB][16], const float (&pmin)[QB][4],
                                              int Tprev, float (&tau_r)[QB], int lanes) {
    if (__builtin_expect(lanes >= MERGE_MIN_LANES, 0)) screen_insert_merge<ITEMS, QB>(C, dv, pmin, Tprev, tau_r);  // laid out of line
    else screen_insert_serial<ITEMS, QB>(C, dv, pmin, Tprev, tau_r);
}

// reduced screening values of one quarter (4 rows) of a finished tile
template <int QB>
__device__ __forceinline__ void sform_part(const SCtx<QB>& C, const f32x16 (&acc)[QB], const f32x4 (&yn)[4], int g,
                                           float (&dv)[QB][16], float (&pmin)[QB][4]) {
    const f32x4 y4 = yn[g];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dv[qb][4 * g + e] = __builtin_fmaf(C.m2s, acc[qb][4 * g + e], y4[e]);
        pmin[qb][g] = fminf(fminf(dv[qb][4 * g], dv[qb][4 * g + 1]), fminf(dv[qb][4 * g + 2], dv[qb][4 * g + 3]));
    }
}

// number of lanes with at least one candidate at or below their threshold
template <int QB>
__device__ __forceinline__ int survivor_lanes(const float (&pmin)[QB][4], const float (&tau_r)[QB]) {
    bool hit = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
        hit |= (fminf(fminf(pmin[qb][0], pmin[qb][1]), fminf(pmin[qb][2], pmin[qb][3])) <= tau_r[qb]);
    return __popcll(__ballot(hit));
}

// h.h' + h.l' + l.h' of one K-slice into each query block's accumulator chain (the fp32 accumulation of all 3*D
// products, in whatever order, is inside screen_band's worst-case bound).  With QB = 2 the two chains alternate,
// so consecutive MFMAs are independent and share the A fragment.
#define TDR_MMA3(AH, AL, S)                                                                             \
    do {                                                                                                \
        _Pragma("unroll") for (int qb = 0; qb < QB; ++qb)                                               \
            acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, bh[qb][S], acc[qb], 0, 0, 0);          \
        if (TERMS == 3) {                                                                               \
            _Pragma("unroll") for (int qb = 0; qb < QB; ++qb)                                           \
                acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, bl[qb][S], acc[qb], 0, 0, 0);      \
            _Pragma("unroll") for (int qb = 0; qb < QB; ++qb)                                           \
                acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, bh[qb][S], acc[qb], 0, 0, 0);      \
        }                                                                                               \
    } while (0)
// LDS tile: TERMS == 3 keeps the {h, l} block pairs of a slice adjacent (block 2s, 2s+1); TERMS == 1 stages the h
// blocks only, compacted (block s)
#define TDR_AH(S) TDR_LDA(ap + ((TERMS == 3 ? 2 : 1) * (S)) * 1024)
#define TDR_AL(S) (TERMS == 3 ? TDR_LDA(ap + (2 * (S) + 1) * 1024) : f16x8{})
#define TDR_LDA(ptr) (*reinterpret_cast<const f16x8*>(ptr))

// One tile step: multiply tile T (A fragments from LDS) into acc and finish tile T-1 out of `prev` between the
// MFMA groups.  Slices are processed in double-buffered groups of GS.
template <int KS, int ITEMS, int QB, int TERMS, bool HAVE_PREV, bool LAZY>
__device__ __forceinline__ void stile_step(const SCtx<QB>& C, const char* __restrict__ img, const f16x8 (&bh)[QB][KS],
                                           const f16x8 (&bl)[QB][KS], f32x16 (&acc)[QB], const f32x16 (&prev)[QB],
                                           const float* ynp_prev, int Tprev, float (&tau_r)[QB], LazyState& S) {
    constexpr int GS = (KS >= 2) ? 2 : 1, NG = KS / GS;
    constexpr int PPG = (4 + NG - 1) / NG;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[qb][r] = 0.f;
    float dv[QB][16];
    float pmin[QB][4];
    // norms of the previous tile's rows first: LDS returns in order, so everything issued after them (the A
    // fragment prefetches) may stay in flight while the epilogue consumes them
    f32x4 yn[4];
    if (HAVE_PREV) {
#pragma unroll
        for (int g = 0; g < 4; ++g) yn[g] = *reinterpret_cast<const f32x4*>(ynp_prev + 8 * g);
    }
    const char* ap = img + C.lane * 16;
    f16x8 ah0[GS], al0[GS], ah1[GS], al1[GS];
#pragma unroll
    for (int u = 0; u < GS; ++u) {
        ah0[u] = TDR_AH(u);
        al0[u] = TDR_AL(u);
    }
    int part = 0;
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
        if (g + 1 < NG) {
#pragma unroll
            for (int u = 0; u < GS; ++u) {
                ah1[u] = TDR_AH((g + 1) * GS + u);
                al1[u] = TDR_AL((g + 1) * GS + u);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (HAVE_PREV) {
#pragma unroll
            for (int pp = 0; pp < PPG; ++pp)
                if (part + pp < 4) sform_part<QB>(C, prev, yn, part + pp, dv, pmin);
        }
        part += PPG;
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            const int s = g * GS + u;
            TDR_MMA3(ah0[u], al0[u], s);
        }
        if (g + 1 < NG) {
            if (g + 2 < NG) {
#pragma unroll
                for (int u = 0; u < GS; ++u) {
                    ah0[u] = TDR_AH((g + 2) * GS + u);
                    al0[u] = TDR_AL((g + 2) * GS + u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (HAVE_PREV) {
#pragma unroll
                for (int pp = 0; pp < PPG; ++pp)
                    if (part + pp < 4) sform_part<QB>(C, prev, yn, part + pp, dv, pmin);
            }
            part += PPG;
#pragma unroll
            for (int u = 0; u < GS; ++u) {
                const int s = (g + 1) * GS + u;
                TDR_MMA3(ah1[u], al1[u], s);
            }
        }
    }
    if (HAVE_PREV) {
        const int lanes = survivor_lanes<QB>(pmin, tau_r);
        TDR_STAT_ADD(5, 1);
#ifdef TDR_SCREEN_STATS
        const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
        if (lanes) {
            if constexpr (LAZY) screen_append_lazy<ITEMS>(C, dv, Tprev, tau_r, S);
            else screen_insert<ITEMS, QB>(C, dv, pmin, Tprev, tau_r, lanes);
        }
#ifdef TDR_SCREEN_STATS
        TDR_STAT_ADD(1, __builtin_readcyclecounter() - ts0);
#endif
    }
}

template <int ITEMS, int QB, bool LAZY>
__device__ __forceinline__ void stile_drain(const SCtx<QB>& C, const f32x16 (&prev)[QB], const float* ynp_prev, int Tprev,
                                            float (&tau_r)[QB], LazyState& S) {
    float dv[QB][16];
    float pmin[QB][4];
    f32x4 yn[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) yn[g] = *reinterpret_cast<const f32x4*>(ynp_prev + 8 * g);
#pragma unroll
    for (int g = 0; g < 4; ++g) sform_part<QB>(C, prev, yn, g, dv, pmin);
    const int lanes = survivor_lanes<QB>(pmin, tau_r);
#ifdef TDR_SCREEN_STATS
    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
    if (lanes) {
        if constexpr (LAZY) screen_append_lazy<ITEMS>(C, dv, Tprev, tau_r, S);
        else screen_insert<ITEMS, QB>(C, dv, pmin, Tprev, tau_r, lanes);
    }
#ifdef TDR_SCREEN_STATS
    TDR_STAT_ADD(1, __builtin_readcyclecounter() - ts0);
#endif
}

// Workgroup = 4 wavefronts x QB query blocks of 32.  QB = 1: 128 queries, two workgroups per CU (two wavefronts
// per SIMD) when the lists fit 80 KiB.  QB = 2: 256 queries, one workgroup per CU, one wavefront per SIMD driving
// two MFMA chains off the same A fragments -- half the LDS reads, LDS-DMA instructions and barriers per matrix
// instruction.
template <int KS, int ITEMS, int QB, int TERMS, bool LAZY = false>
__global__ __launch_bounds__(256, (KS <= 8 && QB == 1 && ITEMS == 1) ? 2 : 1) void knn_screen_kernel(const ScreenParams P) {
    static_assert(!LAZY || QB == 1, "lazy buffers: one query tile per wavefront");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NW = 4;
    constexpr int IMG_B = KS * 2048;                 // bytes of the fragment blocks of one tile image in HBM
    constexpr int LDS_B = KS * 1024 * (TERMS == 3 ? 2 : 1);  // bytes of the staged copy (TERMS == 1: h blocks only)
    constexpr int TILE_F = KS * 512 + 64;            // floats per tile image in HBM
    constexpr int NBLK = 2 * KS;                     // 1-KiB blocks per tile
    char* tile0 = smem_raw;
    char* tile1 = tile0 + LDS_B;
    float* nring = reinterpret_cast<float*>(tile1 + LDS_B);              // [4 slots][64 floats]
    uint64_t* keys_all = reinterpret_cast<uint64_t*>(nring + 4 * 64);    // [NW][QB][32][L]
    const int Ln = P.L;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane & 31, h = lane >> 5;
    uint64_t* keys = keys_all + (size_t)wave * QB * Ln * 32;
#ifdef TDR_SCREEN_STATS
    const unsigned long long t_kernel0 = __builtin_readcyclecounter();
    unsigned long long t_bar = 0;
#define TDR_SYNC() do { const unsigned long long b0_ = __builtin_readcyclecounter(); __syncthreads(); t_bar += __builtin_readcyclecounter() - b0_; } while (0)
#else
#define TDR_SYNC() __syncthreads()
#endif

    const int64_t n_qtiles = (P.nq + 31) / 32;
    const int64_t qt0 = (((int64_t)blockIdx.x + P.batch0) * NW + wave) * QB;
    const bool wave_active = qt0 < n_qtiles;

    const int se = scale_exp(P.meta[0]);
    const float ymax2 = __uint_as_float(P.meta[1]);
    SCtx<QB> C;
    C.P = &P; C.keys = keys; C.lane = lane; C.q = q; C.h = h; C.qt0 = qt0;
    C.m2s = -2.0f * pow2f(-2 * se);

    f16x8 bh[QB][KS], bl[QB][KS];
    float tau_r[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int64_t qt = qt0 + qb;
        const bool blk_active = qt < n_qtiles;
        if (blk_active) {
            const char* qimg = reinterpret_cast<const char*>(P.qp + (size_t)qt * TILE_F);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bh[qb][s] = *reinterpret_cast<const f16x8*>(qimg + (2 * s) * 1024 + lane * 16);
                if (TERMS == 3) bl[qb][s] = *reinterpret_cast<const f16x8*>(qimg + (2 * s + 1) * 1024 + lane * 16);
                else bl[qb][s] = f16x8{};
            }
            C.xn[qb] = reinterpret_cast<const float*>(qimg + IMG_B)[q];
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) { bh[qb][s][e] = (_Float16)0.f; bl[qb][s][e] = (_Float16)0.f; }
            C.xn[qb] = 0.f;
        }
        // rows beyond nq and the padding rows of a cluster-sorted image carry +inf norms: not queries
        const bool lane_valid = blk_active && (qt * 32 + q < P.nq) && (C.xn[qb] < __builtin_inff());
        if (!lane_valid) C.xn[qb] = 0.f;
        C.band[qb] = screen_band(C.xn[qb], ymax2, P.dpad, se, TERMS);
        tau_r[qb] = lane_valid ? __builtin_inff() : -__builtin_inff();
    }
    for (int p = lane; p < QB * Ln * 32; p += 64) keys[p] = KEY_SENTINEL;
    LazyState S = {0, 0, 0};

    const int split = blockIdx.y;
    int t_begin = split * P.tiles_per_split;  // also the origin of the tile / norm ring indices of the current range
    int t_end = t_begin + P.tiles_per_split;
    if (t_end > P.n_db_tiles) t_end = P.n_db_tiles;

    auto stage = [&](int T) {
        const int rel = T - t_begin;
        const float* src = P.yp + (size_t)T * TILE_F;
        char* dst = (rel & 1) ? tile1 : tile0;
        constexpr int NSTG = (TERMS == 3) ? NBLK : KS;  // 1-KiB pieces to stage
#pragma unroll
        for (int t = 0; t < NSTG; t += NW) {
            const int blk = t + wave;
            const int sblk = (TERMS == 3) ? blk : 2 * blk;  // source block: every block, or the h block of slice blk
            if (blk < NSTG)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + sblk * 256 + lane * 4), (lptr_t)(dst + blk * 1024), 16, 0, 0);
        }
        if (wave == NW - 1)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + NBLK * 256 + lane), (lptr_t)(nring + (rel & 3) * 64), 4, 0, 0);
    };
    f32x16 accA[QB], accB[QB];
    // software-pipelined scan of the database tiles [r_begin, r_end): stage(T+1) flies while tile T multiplies and
    // tile T-1 is filtered; the ring indices are relative to t_begin (set to r_begin by the caller)
#define TDR_YN(Tx) (nring + (((Tx) - t_begin) & 3) * 64 + 4 * h)
    auto scan_range = [&](int r_begin, int r_end) {
        if (r_begin >= r_end) return;
        t_begin = r_begin;
        stage(r_begin);
        TDR_SYNC();
        int T = r_begin;
        {
            if (T + 1 < r_end) stage(T + 1);
            if (wave_active) stile_step<KS, ITEMS, QB, TERMS, false, LAZY>(C, tile0, bh, bl, accA, accA, nring, T, tau_r, S);
            TDR_SYNC();
            ++T;
        }
        // ONE copy of the steady-state step (the kernel is ~100 KB of code against a 64 KB instruction cache): the
        // finished tile always sits in accA, the running one in accB, swapped by 16 register moves per tile
        while (T < r_end) {
            if (T + 1 < r_end) stage(T + 1);
            const char* img = ((T - t_begin) & 1) ? tile1 : tile0;
#ifdef TDR_SCREEN_STATS
            const unsigned long long tstep0 = __builtin_readcyclecounter();
#endif
            if (wave_active) stile_step<KS, ITEMS, QB, TERMS, true, LAZY>(C, img, bh, bl, accB, accA, TDR_YN(T - 1), T - 1, tau_r, S);
#ifdef TDR_SCREEN_STATS
            TDR_STAT_ADD(7, __builtin_readcyclecounter() - tstep0);
#endif
            TDR_SYNC();
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) accA[qb] = accB[qb];
            ++T;
        }
        if (wave_active) stile_drain<ITEMS, QB, LAZY>(C, accA, TDR_YN(r_end - 1), r_end - 1, tau_r, S);
        TDR_SYNC();  // the last norm-ring slot / tile buffers may be restaged by the next range
    };

    {
        // Range driver (ONE call site of scan_range: the scan body is most of the kernel's code).  Without cluster tables:
        // the slice [t_begin, t_end).  With them, cluster-bound pruning: points are sorted by cluster and clusters start
        // on tile boundaries, so a wavefront's 32 queries lie in ONE cluster ball B(c_w, R_w).  Every member y of
        // cluster c satisfies |x - y| >= |c_w - c_c| - R_w - R_c, hence its screening value a >= lb - E (E <= band / 2).
        // Clusters are visited by increasing centre distance from the first wavefront's cluster (tightens the
        // thresholds early); one whose bound exceeds the largest current threshold of the workgroup can never
        // contribute a candidate and is skipped -- thresholds only decrease, so the decision stays valid.  Exactness
        // does not depend on the clustering quality, only the amount of skipped work does.
        const bool pruned = P.tile_cluster != nullptr;
        float* wred = reinterpret_cast<float*>(keys_all + (size_t)NW * QB * 32 * Ln);  // 16 floats behind the lists
        unsigned long long* wmask = reinterpret_cast<unsigned long long*>(wred + 8);   // 4 ballots
        int cw[NW];
        int64_t qtile[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int64_t qtw = (((int64_t)blockIdx.x + P.batch0) * NW + w) * QB;
            cw[w] = (pruned && qtw < n_qtiles) ? P.tile_cluster[qtw] : -1;
            qtile[w] = qtw;
        }
        float bmax = 0.f;  // band of the valid lanes only (invalid lanes were given ||x||^2 = 0)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) bmax = fmaxf(bmax, C.band[qb]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, o, 64));
        // largest threshold of the workgroup in screening (a) units (tau <= tau_r + ||x||^2) and its largest band
        auto wg_threshold = [&](float& tau_wg, float& band_wg) {
            float tmax = -__builtin_inff();
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) tmax = fmaxf(tmax, tau_r[qb] + C.xn[qb]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
            if (lane == 0) { wred[wave] = wave_active ? tmax : -__builtin_inff(); wred[4 + wave] = wave_active ? bmax : 0.f; }
            __syncthreads();
            tau_wg = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
            band_wg = fmaxf(fmaxf(wred[4], wred[5]), fmaxf(wred[6], wred[7]));
            __syncthreads();
        };
        float tau_wg = 0.f, band_wg = 0.f;
        int idx0 = 0;
        bool more = true;
        // approximate mode (max_visit > 0, the IVF search of distance/faiss.py:331-349 on the cluster index): clusters are
        // taken by increasing distance of their centre to the NEAREST of the wavefronts' own centres: the own clusters
        // (distance 0) first, then max_visit - 1 others.  The candidates are the first max_visit entries of every wavefront's
        // visiting order (their union contains the max_visit nearest by that key); "next" = the smallest (key, cluster)
        // above the last one taken, so no visited set is kept.  A cluster the exact bound excludes takes its probe but is
        // not scanned.
        int visited = 0;
        unsigned long long prev = 0ull;  // (key bits << 32 | cluster) + 1 of the last cluster taken
        // exact mode: the distinct own clusters of the wavefronts, a cursor into the visiting order of each, whose turn it is
        int own[NW], idxj[NW], n_own = 0, rr = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { own[w] = 0; idxj[w] = 0; }
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            bool fresh = cw[w] >= 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) fresh = fresh && !(w2 < w && cw[w2] == cw[w]);
            if (fresh) {
#pragma unroll
                for (int t = 0; t < NW; ++t)
                    if (t == n_own) own[t] = cw[w];
                ++n_own;
            }
        }
        unsigned* vis = reinterpret_cast<unsigned*>(wmask + 4);     // 4096 bits: clusters scanned by this workgroup
        if (pruned) {
            if (tid < 128) vis[tid] = 0u;
            __syncthreads();
        }
        while (more) {
            int rb = t_begin, re = t_end;
            int c_found = -1;    // exact pruned mode: the cluster about to be scanned
            if (!pruned) {
                more = false;
            } else {
                bool found = false;
                wg_threshold(tau_wg, band_wg);  // thresholds only decrease: later tests get sharper
                if (P.max_visit > 0) {
                    unsigned long long best = ~0ull;
                    const int ncand = P.max_visit < P.n_clusters ? P.max_visit : P.n_clusters;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        if (cw[w] < 0) continue;
                        for (int i = tid; i < ncand; i += 256) {
                            const int c = P.clus_order[(size_t)cw[w] * P.n_clusters + i];
                            float key = __builtin_inff();
#pragma unroll
                            for (int w2 = 0; w2 < NW; ++w2) {
                                if (cw[w2] < 0) continue;
                                key = fminf(key, P.clus_dist[(size_t)cw[w2] * P.n_clusters + c]);
                            }
                            const unsigned long long kc = (((unsigned long long)__float_as_uint(fmaxf(key, 0.f)) << 32) | (uint32_t)c) + 1ull;
                            if (kc > prev && kc < best) best = kc;
                        }
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const unsigned long long other = __shfl_xor(best, o, 64);
                        best = other < best ? other : best;
                    }
                    if (lane == 0) wmask[wave] = best;
                    __syncthreads();
#pragma unroll
                    for (int w = 0; w < NW; ++w) best = wmask[w] < best ? wmask[w] : best;
                    __syncthreads();
                    if (best == ~0ull) break;
                    // the wavefronts' own clusters (key 0) are always scanned -- a query's own list is Faiss's nprobe = 1 --
                    // and count as ONE scan together; max_visit - 1 further clusters follow
                    const bool own_cluster = ((best - 1ull) >> 32) == 0ull;
                    if (!own_cluster && visited >= P.max_visit - 1) break;
                    if (own_cluster) --visited;  // undone below by the common increment
                    prev = best;
                    const int c = (int)(uint32_t)((best - 1ull) & 0xffffffffull);
                    // The probe set is a function of the index alone (the lists by (centre distance to the nearest own centre,
                    // id)): a list the exact bound excludes -- no member can enter any band of this workgroup -- still takes its
                    // probe, it is just not scanned (scanning it would change nothing).  The result is therefore exactly "top-k
                    // over the rows of the probed lists", which oracle/ref_torch.py:ivf_search restates on the CPU.
                    float lbc = __builtin_inff();
                    {
                        const float rc = P.clus_radius[c];
#pragma unroll
                        for (int w2 = 0; w2 < NW; ++w2) {
                            if (cw[w2] < 0) continue;
                            const float g = P.clus_dist[(size_t)cw[w2] * P.n_clusters + c] - P.clus_radius[cw[w2]] - rc;
                            lbc = fminf(lbc, g > 0.f ? g * g : 0.f);
                        }
                    }
                    const bool excluded = lbc * 0.9999f - band_wg > tau_wg;
                    rb = excluded ? 0 : P.clus_tile_begin[c];
                    re = excluded ? 0 : P.clus_tile_begin[c + 1];
                    found = true;
                }
                // Exact mode.  A workgroup's four query tiles may lie in up to four clusters, and the layout order of the
                // clusters says nothing about where they are in space: walking only the visiting order of the first
                // wavefront's cluster, a tile of another cluster met its own neighbourhood after hundreds of clusters, and until
                // then its thresholds -- those of arbitrary rows -- let nothing be pruned for the whole workgroup (N = 700k in
                // 1000 blobs of 22 tiles: a fifth of the workgroups straddle two blobs, 47 ms against 18 ms at N = 1M where
                // blobs are 8 workgroups each).  So the visiting orders of ALL the distinct own clusters are walked in turn
                // (each starts with the cluster itself), a bitmap in LDS keeps a cluster from being scanned twice.
                while (!found && P.max_visit <= 0) {
                    int j = -1;
#pragma unroll
                    for (int t = 0; t < NW; ++t) {
                        const int jj = (rr + t) % NW;
                        if (j < 0 && jj < n_own && idxj[jj] < P.n_clusters) j = jj;
                    }
                    if (j < 0) break;
                    rr = (j + 1) % NW;
                    int cj = own[0], ij = idxj[0];
#pragma unroll
                    for (int t = 1; t < NW; ++t)
                        if (t == j) { cj = own[t]; ij = idxj[t]; }
                    // 256 clusters of the visiting order are tested at once, one per thread, against the current threshold
                    const int idx = ij + tid;
                    bool survive = false;
                    if (idx < P.n_clusters) {
                        const int c = P.clus_order[(size_t)cj * P.n_clusters + idx];
                        const float rc = P.clus_radius[c];
                        float lb = __builtin_inff();
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            if (cw[w] < 0) continue;
                            // ball to ball: |x - y| >= |c_w - c_c| - R_w - R_c; with the per-tile table, from the tile's own
                            // rows: |x - y| >= min_x |x - c_c| - R_c (no R_w, and |x - c_c| ~ sqrt(|c_w - c_c|^2 + |x - c_w|^2)
                            // in high dimension: prunes where the balls themselves overlap)
                            float g = P.clus_dist[(size_t)cw[w] * P.n_clusters + c] - P.clus_radius[cw[w]] - rc;
                            if (P.tile_cdist) {
#pragma unroll
                                for (int qb = 0; qb < QB; ++qb)
                                    g = fmaxf(g, P.tile_cdist[(size_t)(qtile[w] + qb) * P.n_clusters + c] - rc);
                            }
                            lb = fminf(lb, g > 0.f ? g * g : 0.f);
                        }
                        // pruned only when NO member can enter any band; scanned already: not again (duplicates in the lists)
                        survive = !((vis[c >> 5] >> (c & 31)) & 1u) && !(lb * 0.9999f - band_wg > tau_wg);
                    }
                    const unsigned long long m = __ballot(survive);
                    if (lane == 0) wmask[wave] = m;
                    __syncthreads();
                    int first = -1;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const unsigned long long mw = wmask[w];
                        if (first < 0 && mw) first = 64 * w + __builtin_ctzll(mw);
                    }
                    __syncthreads();
                    const int adv = first < 0 ? 256 : first + 1;
#pragma unroll
                    for (int t = 0; t < NW; ++t)
                        if (t == j) idxj[t] += adv;
                    if (first < 0) continue;
                    const int c = P.clus_order[(size_t)cj * P.n_clusters + ij + first];
                    if (tid == 0) vis[c >> 5] |= 1u << (c & 31);     // read again only after the barriers of the scan below
                    rb = P.clus_tile_begin[c];
                    re = P.clus_tile_begin[c + 1];
                    c_found = c;
                    found = true;
                }
                if (!found) break;
                ++visited;
            }
            if constexpr (LAZY) {
                // The lazy buffers refresh a threshold only when a buffer fills up, and the test above read them as they were.  Before
                // a cluster is actually scanned: bring the thresholds of every wavefront up to date and test THIS cluster again -- a
                // workgroup whose stale thresholds already exclude everything (the headline data after its own cluster) never pays
                // for the refresh.
                if (pruned && P.max_visit <= 0 && c_found >= 0) {
                    const bool stale = wave_active && __any(!S.lost && S.cnt >= P.k && (S.fresh >= 8 || (S.fresh > 0 && tau_r[0] == __builtin_inff())));
                    if (lane == 0) wmask[wave] = stale ? 1ull : 0ull;
                    __syncthreads();
                    const bool any_stale = (wmask[0] | wmask[1] | wmask[2] | wmask[3]) != 0ull;
                    __syncthreads();
                    if (any_stale) {
                        if (wave_active) lazy_refresh<ITEMS>(C, tau_r, S, 8);
                        wg_threshold(tau_wg, band_wg);
                        const float rc = P.clus_radius[c_found];
                        float lb = __builtin_inff();
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            if (cw[w] < 0) continue;
                            float g = P.clus_dist[(size_t)cw[w] * P.n_clusters + c_found] - P.clus_radius[cw[w]] - rc;
                            if (P.tile_cdist) {
#pragma unroll
                                for (int qb = 0; qb < QB; ++qb)
                                    g = fmaxf(g, P.tile_cdist[(size_t)(qtile[w] + qb) * P.n_clusters + c_found] - rc);
                            }
                            lb = fminf(lb, g > 0.f ? g * g : 0.f);
                        }
                        if (lb * 0.9999f - band_wg > tau_wg) { rb = 0; re = 0; }   // excluded after all (its bit in `vis` stays: thresholds only fall)
                    }
                }
            }
            scan_range(rb, re);
        }
    }
#undef TDR_YN
#ifdef TDR_SCREEN_STATS
    if (lane == 0 && wave_active) {
        atomicAdd(&g_screen_stats[0][blockIdx.x & 63], __builtin_readcyclecounter() - t_kernel0);
        atomicAdd(&g_screen_stats[6][blockIdx.x & 63], t_bar);
    }
#endif
#undef TDR_SYNC

    if (wave_active) {
        const int Lo = P.L_out;
        if constexpr (LAZY) {
            if (Lo < Ln) {
                // short output lists (the pilots: 64 slices x 512 queries meet in the rescoring kernel's LDS): one last compaction
                // leaves k + the band's population in every buffer; more than the list takes = the sorted list would have overflowed
                lazy_refresh<ITEMS>(C, tau_r, S, 1, true);
                if (S.cnt >= Lo && S.cnt >= P.k) S.lost = 1;     // the state in which a sorted list of Lo entries is full inside its band
            }
        }
        for (int jq = 0; jq < 32 * QB; ++jq) {
            const int64_t qi = qt0 * 32 + jq;
            if (qi >= P.nq) break;
            if constexpr (LAZY) {
                // the buffer's entries in arrival order, sentinels behind them (the rescoring kernel ranks by counting)
                const int nj = __builtin_amdgcn_readlane(S.cnt, jq);
                for (int p = lane; p < Lo; p += 64)
                    P.cand[((size_t)split * P.nq + qi) * Lo + p] = p < nj ? keys[(size_t)jq * Ln + p] : KEY_SENTINEL;
            } else {
                for (int p = lane; p < Ln; p += 64)
                    P.cand[((size_t)split * P.nq + qi) * Ln + p] = keys[(size_t)jq * Ln + p];
            }
        }
        if constexpr (LAZY) {
            const int64_t qi = qt0 * 32 + q;
            if (h == 0 && qi < P.nq && S.lost && P.lost) P.lost[qi] = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Rescoring: one wavefront per query.  Candidates (all splits) -> global k-th screening value -> band ->
// exact fp32 distance of every candidate inside the band (reference arithmetic) -> rank by (distance, index).
// A query whose spare list slots overflowed in any split is flagged for the exact one-stage kernel.
// ---------------------------------------------------------------------------------------------------------
// The rescoring kernel packs the candidates inside the band before it evaluates them: room for all of them when the lists are
// short, for 512 otherwise (a pilot over 64 database slices brings ~4000 list entries per query; a band that holds more than 512
// of them is far beyond every list length a launch could keep -- the query is flagged)
__host__ __device__ __forceinline__ int rescore_band_cap(int total) { return total < 512 ? total : 512; }

struct RescoreParams {
    const uint64_t* cand;  // (n_splits, nq, L)
    const float* Xq;       // (nq, d) row-major queries, row stride ldq
    const float* Y;        // (n_db, d) row-major database, row stride ldy
    const float* norms_q;  // (nq) reference-order squared norms
    const float* norms_y;  // (n_db)
    const uint32_t* meta;
    int64_t nq, ldq, ldy;
    int d, dpad, k, L, n_splits, metric, terms;
    int predict_unsplit;   // pilot runs: also flag queries whose band holds >= pred_L candidates over ALL slices
    int pred_L;            // list length the prediction is made for (the launch's own L, or the longer lists of the threshold scan)
    int pred_terms;        // 0: the prediction counts the launch's own band; 2: the (wider) band of the two-term split h.h' + h.l', counted
                           // on a three-term pilot's near-exact screening values -- the threshold scan's two-term tier is chosen by it
    const int32_t* lost;   // optional (nq): 1 = the threshold scan dropped candidates of this query (buffer capacity)
    int unsorted;          // 1: the lists are the lazy buffers of the pruned scan (any order, overflow reported through `lost`)
    const int32_t* row_map; // screening index -> source row (cluster-sorted search), NULL = identity
    int64_t q_begin, q_end; // screening positions handled by this launch
    float* out_d;
    int32_t* out_i;
    int32_t* flags;        // (nq) 1 = overflow
    int32_t* n_flagged;    // device counter
};

__global__ __launch_bounds__(256) void knn_rescore_kernel(const RescoreParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = P.n_splits * P.L;
    const int dq = (P.d + 3) & ~3;
    // per-wave LDS: approx keys [total], exact keys [total], query row [dq], scalar
    const int ecap = rescore_band_cap(total);    // packed in-band candidates a wavefront can hold (more: the query is flagged)
    const size_t per_wave = (size_t)(total + ecap) * sizeof(uint64_t) + (size_t)dq * sizeof(float) + 16;
    char* base = smem_raw + (size_t)wave * per_wave;
    uint64_t* ak = reinterpret_cast<uint64_t*>(base);
    uint64_t* ek = ak + total;
    float* xq = reinterpret_cast<float*>(ek + ecap);
    uint32_t* sc = reinterpret_cast<uint32_t*>(xq + dq);
    const int64_t qi = P.q_begin + (int64_t)blockIdx.x * 4 + wave;
    if (qi >= P.q_end) return;  // no block-level barrier below: wavefronts are independent
    // cluster-sorted search: screening position -> source row (outputs go to the source row, keys carry source indices
    // so that the canonical (distance, index) order is the caller's)
    const int64_t qs = P.row_map ? (int64_t)P.row_map[qi] : qi;
    if (qs < 0) return;  // padding row

    for (int p = lane; p < total; p += 64) {
        const int s = p / P.L, r = p - s * P.L;
        ak[p] = P.cand[((size_t)s * P.nq + qi) * P.L + r];
    }
    for (int c = lane; c < dq; c += 64) xq[c] = (c < P.d) ? P.Xq[(size_t)qs * P.ldq + c] : 0.f;
    if (lane == 0) sc[0] = 0xFF800000u;
    const float nx = P.norms_q[qs];
    const int se = scale_exp(P.meta[0]);
    const float band = screen_band(nx, __uint_as_float(P.meta[1]), P.dpad, se, P.terms);
    const float band_pred = P.pred_terms ? screen_band(nx, __uint_as_float(P.meta[1]), P.dpad, se, P.pred_terms) : band;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // overflow test per split: list full and its last entry inside that split's band
    bool ovf = false;
    for (int s = lane; s < (P.unsorted ? 0 : P.n_splits); s += 64) {
        const uint64_t kl = ak[s * P.L + P.L - 1], kk = ak[s * P.L + P.k - 1];
        if (kl != KEY_SENTINEL && P.L > P.k) ovf |= u2f((uint32_t)(kl >> 32)) <= u2f((uint32_t)(kk >> 32)) + band;
        if (P.L == P.k) ovf = true;
    }
    const bool any_ovf = __any(ovf);

    // global k-th smallest screening VALUE: the smallest bit image with at least k entries at or below it, by bisection (a ballot
    // and a popcount per 64 entries and step).  Rounds 1-5 ranked every key against every other -- total^2 / 64 compares per
    // lane: 1.2-2 ms for the 512 queries of a pilot (32 slices x L entries each, on the critical path of the kNN build) and, with
    // the 126-entry buffers of the lazy pruned scan, 4.3 ms at N = 1M.
    int nv = 0;
    for (int p0 = 0; p0 < total; p0 += 64) {
        const int p = p0 + lane;
        nv += __popcll(__ballot(p < total && ak[p] != KEY_SENTINEL));
    }
    uint32_t kth = 0xFF800000u;      // fewer than k candidates: +inf, everything is inside the band
    if (nv >= P.k) {
        // the range of the finite images (their smallest and largest: DPP reductions), then the bisection.  A pilot needs the k-th
        // value itself (its flags count the band's population); any other launch may stop at a bound that holds up to 4 entries
        // more than k -- a few more candidates get their exact distance in the same pass, the result is the same
        uint32_t vmin = 0xffffffffu, vmax = 0u;
        for (int p0 = 0; p0 < total; p0 += 64) {
            const int p = p0 + lane;
            const uint32_t v = (p < total) ? (uint32_t)(ak[p] >> 32) : 0xFF800000u;
            vmin = v < vmin ? v : vmin;
            if (v < 0xFF800000u) vmax = v > vmax ? v : vmax;
        }
        uint32_t lo = wave_extreme_u32<false>(vmin), hi = wave_extreme_u32<true>(vmax);
        const int slack = P.predict_unsplit ? 0 : 4;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            int c = 0;
            for (int p0 = 0; p0 < total; p0 += 64) {
                const int p = p0 + lane;
                c += __popcll(__ballot(p < total && (uint32_t)(ak[p] >> 32) <= mid));
            }
            if (c >= P.k) {
                hi = mid;
                if (c <= P.k + slack) break;
            } else {
                lo = mid + 1u;
            }
        }
        kth = hi;
    }
    const float thr = u2f(kth) + band;
    const float thr_pred = u2f(kth) + band_pred;

    // the candidates inside the band, packed: ek[0 .. m)
    int m = 0, in_pred = 0;
    for (int p0 = 0; p0 < total; p0 += 64) {
        const int p = p0 + lane;
        const uint64_t key = (p < total) ? ak[p] : KEY_SENTINEL;
        const float av = u2f((uint32_t)(key >> 32));
        const bool inb = key != KEY_SENTINEL && av <= thr;
        in_pred += __popcll(__ballot(key != KEY_SENTINEL && av <= thr_pred));
        const unsigned long long mk = __ballot(inb);
        const int pos = m + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0));
        if (inb && pos < ecap) ek[pos] = key;
        m += __popcll(mk);
    }
    const int in_band = m;
    const bool band_overflow = m > ecap;
    if (band_overflow) m = ecap;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // their exact distances (reference arithmetic), one candidate per lane: ak[0 .. m) (the screening keys are not needed any more)
    for (int p0 = 0; p0 < m; p0 += 64) {
        const int p = p0 + lane;
        if (p < m) {
            const uint64_t mine = ek[p];
            const uint32_t jp = (uint32_t)(mine & 0xffffffffu);
            const uint32_t j = P.row_map ? (uint32_t)P.row_map[jp] : jp;
            const float* yr = P.Y + (size_t)j * P.ldy;
            float acc = 0.f;
            int c = 0;
            if ((P.ldy & 3) == 0 && ((uintptr_t)P.Y & 15) == 0) {
                for (; c + 4 <= P.d; c += 4) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(yr + c);
                    acc = __builtin_fmaf(xq[c], yv[0], acc);
                    acc = __builtin_fmaf(xq[c + 1], yv[1], acc);
                    acc = __builtin_fmaf(xq[c + 2], yv[2], acc);
                    acc = __builtin_fmaf(xq[c + 3], yv[3], acc);
                }
            }
            for (; c < P.d; ++c) acc = __builtin_fmaf(xq[c], yr[c], acc);
            const float cval = __builtin_fmaf(-2.0f, acc, __fadd_rn(nx, P.norms_y[j]));
            ak[p] = mkkey(cval, j);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // rank by (distance, index) among the m rescored candidates
    for (int p0 = 0; p0 < m; p0 += 64) {
        const int p = p0 + lane;
        const uint64_t mine = (p < m) ? ak[p] : KEY_SENTINEL;
        int rank = 0;
        for (int pp = 0; pp < m; ++pp) rank += (ak[pp] < mine) ? 1 : 0;
        if (p < m && rank < P.k) {
            float c = u2f((uint32_t)(mine >> 32));
            if (P.metric == 1) c = sqrt_rn(fmaxf(c, 0.f));
            P.out_d[(size_t)qs * P.k + rank] = c;
            P.out_i[(size_t)qs * P.k + rank] = (int32_t)(uint32_t)(mine & 0xffffffffu);
        }
    }
    // A database-sliced launch keeps L entries PER SLICE, so it overflows far less than the unsliced launch of the
    // same search would; a pilot that stands for an unsliced run predicts from the merged band population instead.
    bool flag = any_ovf || band_overflow || (P.predict_unsplit && (P.pred_terms ? in_pred : in_band) >= P.pred_L);
    if (P.lost && P.lost[qi] != 0) flag = true;
    if (lane == 0) {
        P.flags[qs] = flag ? 1 : 0;
        if (flag) atomicAdd(P.n_flagged, 1);
    }
}

static inline int pick_ks(int d) {
    if (d <= 32) return 2;
    if (d <= 64) return 4;
    if (d <= 128) return 8;
    if (d <= 256) return 16;  // 32-KiB tiles and 128 query-fragment VGPRs: one workgroup per CU (one wavefront per SIMD)
    return 0;
}

static size_t screen_lds_bytes(int ks, int L, int qb, int terms) {
    return (size_t)2 * ks * 1024 * (terms == 3 ? 2 : 1) + (size_t)4 * 64 * sizeof(float) +
           (size_t)4 * qb * 32 * L * sizeof(uint64_t) + 128 + 512;  // + the workgroup reduction slots and the scanned-cluster bitmap of the pruned scan
}

// Workgroup shape, number of split terms and list length per tier.  The list holds k entries plus spare slots for
// the candidates inside the error band.
//   tier 0: ONE term (h.h' only): a third of the matrix work and half the staged bytes, band ~2^-10 ||x|| ||y||;
//           the h-only tile leaves room for lists of up to 62 entries with two workgroups per CU.
//   tier 1: three terms, band ~1e-4 ||x|| ||y||, k + ~17..24 spare slots, two workgroups per CU (80 KiB each);
//           larger k: one workgroup per CU, two list entries per lane (L <= 128).
//   tier 2: three terms, one workgroup per CU, up to k + 72 spare slots.
struct ScreenCfg { int qb, L, items, wg_per_cu, terms, lazy; };

static int screen_qb_pref() { return 1; }  // 256 queries per workgroup measured slower (786 vs 759 ms at 1M)

static int max_list_len(int ks, int qb, int terms, size_t budget, int cap) {
    int L = 0;
    while (L + 1 <= cap && screen_lds_bytes(ks, L + 1, qb, terms) <= budget) ++L;
    return L;
}

static ScreenCfg screen_cfg(int ks, int k, int tier) {
    const int spare_min = 8;
    ScreenCfg c = {0, 0, 0, 0, 3, 0};
    if (tier == 0) {
        const int L2 = max_list_len(ks, 1, 1, (ks > 8 ? 160 : 80) * 1024, 64);
        if (k + 16 <= L2) { c.qb = 1; c.L = L2; c.items = 1; c.wg_per_cu = ks > 8 ? 1 : 2; c.terms = 1; }
        return c;
    }
    if (tier == 1 && ks <= 8) {
        if (screen_qb_pref() == 2) {
            const int Lq = max_list_len(ks, 2, 3, 160 * 1024, 64);
            if (k + spare_min <= Lq) { c.qb = 2; c.L = (k + 24 < Lq) ? k + 24 : Lq; c.items = 1; c.wg_per_cu = 1; return c; }
        }
        const int L2 = max_list_len(ks, 1, 3, 80 * 1024, 64);
        if (k + spare_min <= L2) { c.qb = 1; c.L = (k + 24 < L2) ? k + 24 : L2; c.items = 1; c.wg_per_cu = 2; return c; }
    }
    const int L1 = max_list_len(ks, 1, 3, 160 * 1024, 128);
    const int spare = tier == 1 ? 32 : 72;
    if (k + spare_min <= L1) { c.qb = 1; c.L = (k + spare < L1) ? k + spare : L1; c.items = c.L > 64 ? 2 : 1; c.wg_per_cu = 1; return c; }
    return c;
}

// The cluster-pruned scan with lazy buffers (screen_append_lazy): one workgroup per CU, the buffers take what the tiles leave of
// the CU's 160 KiB (126 entries per query at D <= 128, 94 at D <= 256); worth it only while a buffer holds a tile's survivors
// next to the k + band entries a compaction keeps (L >= 2 k, L >= k + 40), otherwise the sorted lists serve the search.
static int g_clustered_lazy = 1;     // tdr_knn_screen_clustered_lists
static ScreenCfg lazy_cfg(int ks, int k, const ScreenCfg& base) {
    ScreenCfg c = base;
    c.lazy = 0;
    if (!g_clustered_lazy || base.L == 0) return c;
    // an ODD number of 8-byte entries per query: the 32 queries of a wavefront store to 32 different bank pairs (126 entries: a
    // row stride of 252 dwords puts them on 16)
    const int Lz = (max_list_len(ks, 1, base.terms, 160 * 1024, 128) - 1) | 1;
    if (Lz < 2 * k || Lz < k + 40 || Lz <= base.L) return c;
    c.qb = 1; c.L = Lz; c.items = 2; c.wg_per_cu = 1; c.lazy = 1;
    return c;
}

static int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    return cus;
}

// database splits (gridDim.y) so that small query counts still fill the chip
static int screen_splits(int64_t nq, int n_db_tiles, const ScreenCfg& c) {
    const int64_t wgs = (nq + 128 * c.qb - 1) / (128 * c.qb);
    const int target = 2 * device_cus() * c.wg_per_cu;
    if (wgs >= target) return 1;
    int64_t s = (target + wgs - 1) / wgs;
    const int64_t max_by_tiles = n_db_tiles / 64 > 0 ? n_db_tiles / 64 : 1;
    if (s > max_by_tiles) s = max_by_tiles;
    // the rescoring kernel keeps 2 x splits x L keys per wavefront in LDS: 4 wavefronts x 16 B x splits x L <= ~144 KiB
    // the rescoring kernel keeps splits x L list entries + up to 512 packed candidates per wavefront in LDS (4 wavefronts, 8 B each)
    const int64_t max_by_lds = (36 * 1024 / 8 - 512) / (int64_t)(c.L > 0 ? c.L : 1);
    if (s > max_by_lds) s = max_by_lds;
    // 64 slices (32 until round 6): a pilot of 512 queries is 4 query groups x 64 slices = 256 workgroups of ~490 tile steps -- its
    // time is the length of that dependent chain (the next tile's load is issued one step ahead), not the amount of work
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return (int)s;
}

template <int KS, int ITEMS, int QB, int TERMS, bool LAZY = false>
static int launch_screen(const ScreenParams& P, int n_wgs, size_t lds, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_screen_kernel<KS, ITEMS, QB, TERMS, LAZY>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((knn_screen_kernel<KS, ITEMS, QB, TERMS, LAZY>), dim3((unsigned)n_wgs, (unsigned)P.n_splits), dim3(256), lds, st, P);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

template <int KS>
static int launch_screen_ks(const ScreenParams& P, const ScreenCfg& c, int n_wgs, size_t lds, hipStream_t st) {
    if (c.lazy) return c.terms == 1 ? launch_screen<KS, 2, 1, 1, true>(P, n_wgs, lds, st) : launch_screen<KS, 2, 1, 3, true>(P, n_wgs, lds, st);
    if (c.terms == 1) return launch_screen<KS, 1, 1, 1>(P, n_wgs, lds, st);
    if constexpr (KS <= 8) {
        if (c.qb == 2) return launch_screen<KS, 1, 2, 3>(P, n_wgs, lds, st);
    }
    if (c.items == 1) return launch_screen<KS, 1, 1, 3>(P, n_wgs, lds, st);
    return launch_screen<KS, 2, 1, 3>(P, n_wgs, lds, st);
}

}  // namespace scr
}  // namespace tdr

using namespace tdr;
using namespace tdr::scr;

extern "C" {

/* 1 when the two-stage (screen + rescore) search supports feature dimension d and neighbour count k. */
int tdr_knn_screen_supported(int d, int k) {
    const int ks = pick_ks(d);
    if (ks == 0 || k < 1) return 0;
    return screen_cfg(ks, k, 1).L > 0 ? 1 : 0;
}

/* Floats of the fp16-split image of n rows of dimension d (0 if unsupported). */
int64_t tdr_packed16_floats(int64_t n, int d) {
    const int ks = pick_ks(d);
    if (ks == 0 || n < 0) return 0;
    return ((n + TILE_ROWS - 1) / TILE_ROWS) * tile16_stride_floats(ks);
}

/* meta (2 x uint32, device, caller-zeroed before the first call): meta[0] = max(meta[0], bits(max |X|)) over the
 * n x d block; when norms != NULL also meta[1] = max(meta[1], bits(max norms[i])).  Call once per point block
 * that will be packed with this meta (queries and database share one scale). */
int tdr_screen_meta_f32(const float* X, int64_t n, int d, int64_t ldx, const float* norms, uint32_t* meta, void* stream) {
    if (!X || !meta || n <= 0 || d <= 0 || ldx < d) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = n * d;
    unsigned grid = (unsigned)((total + 256 * 16 - 1) / (256 * 16));
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    if (ldx == d) hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, st, X, total, meta);
    else hipLaunchKernelGGL(absmax2d_kernel, dim3(grid), dim3(256), 0, st, X, n, d, ldx, meta);
    TDR_CHECK_LAUNCH();
    if (norms) {
        unsigned g2 = (unsigned)((n + 256 * 16 - 1) / (256 * 16));
        if (g2 > 1024) g2 = 1024;
        if (g2 < 1) g2 = 1;
        hipLaunchKernelGGL(absmax_kernel, dim3(g2), dim3(256), 0, st, norms, n, meta + 1);
        TDR_CHECK_LAUNCH();
    }
    return TDR_OK;
}

int tdr_pack16_mapped_f32(const float* X, int64_t n, int d, int64_t ldx, const float* norms, const uint32_t* meta,
                          const int32_t* row_map, float* packed16, void* stream);

/* fp16-split tile images of X; norms = the reference-order squared norms written by tdr_pack_rows_f32. */
int tdr_pack16_f32(const float* X, int64_t n, int d, int64_t ldx, const float* norms, const uint32_t* meta, float* packed16,
                   void* stream) {
    return tdr_pack16_mapped_f32(X, n, d, ldx, norms, meta, nullptr, packed16, stream);
}

/* As tdr_pack16_f32 for a re-ordered, padded image: image row r holds source row row_map[r] of X (norms indexed by
 * source row), -1 = padding row (zero features, +inf norm).  n = number of IMAGE rows. */
int tdr_pack16_mapped_f32(const float* X, int64_t n, int d, int64_t ldx, const float* norms, const uint32_t* meta,
                          const int32_t* row_map, float* packed16, void* stream) {
    if (!X || !norms || !meta || !packed16 || n <= 0 || d <= 0 || ldx < d) return TDR_ERR_BAD_ARG;
    const int ks = pick_ks(d);
    if (ks == 0) return TDR_ERR_UNSUPPORTED;
    const int64_t tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
    const size_t shmem = (size_t)TILE_ROWS * (ks * 16 + 4) * sizeof(float);
    hipLaunchKernelGGL(pack16_kernel, dim3((unsigned)tiles), dim3(256), shmem, (hipStream_t)stream, X, n, d, ldx, ks, norms,
                       meta, row_map, packed16);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

int64_t tdr_knn_screen_workspace_bytes(int64_t nq, int64_t n_db, int d, int k, int tier) {
    const int ks = pick_ks(d);
    if (ks == 0) return 0;
    const ScreenCfg c = screen_cfg(ks, k, tier);
    if (c.L == 0) return 0;
    const int n_db_tiles = (int)((n_db + TILE_ROWS - 1) / TILE_ROWS);
    const int splits = screen_splits(nq, n_db_tiles, c);
    // + one word per query: the lost marks of the launches that keep lazy buffers (pilots)
    return (int64_t)splits * nq * c.L * (int64_t)sizeof(uint64_t) + ((nq * 4 + 15) / 16) * 16;
}

/*
 * Two-stage exact kNN (sqeuclidean / euclidean).  q16 / y16: fp16-split images (tdr_pack16_f32, same meta);
 * Xq / Y: the row-major fp32 blocks they were packed from; norms_q / norms_y: reference-order squared norms.
 * out_d / out_i as tdr_knn_packed_f32.  flags (nq int32): 1 where the screening list overflowed -- those rows
 * of out_d / out_i are NOT valid and must be recomputed with tdr_knn_packed_f32; *n_flagged (device int32,
 * caller-zeroed) counts them.  tier: 0 = one-term screening (h.h' only; cheapest, widest band; TDR_ERR_UNSUPPORTED when
 * k + 16 list slots do not fit), 1 = three-term screening with k + ~17..24 spare slots, 2 = three terms and long
 * lists (up to k + 72 spare slots, one workgroup per CU) for data whose error band holds more candidates.
 * predict_unsplit = 1 (pilot slices): additionally flag queries whose error band holds >= L candidates over all
 * database slices together, i.e. the ones an unsliced launch of the same search would flag.
 */
struct ClusterTables {
    int n_clusters;
    const int32_t* row_map;
    const int32_t* tile_cluster;
    const int32_t* clus_tile_begin;
    const float* clus_radius;
    const float* clus_dist;
    const int32_t* clus_order;
    int max_visit;
    const float* tile_cdist;
};

static int launch_lists_scan(const ScreenParams& P, const ScreenCfg& cfg, int ks, int wgs, hipStream_t st) {
    const size_t lds = screen_lds_bytes(ks, P.L, cfg.qb, cfg.terms);
    switch (ks) {
        case 2: return launch_screen_ks<2>(P, cfg, wgs, lds, st);
        case 4: return launch_screen_ks<4>(P, cfg, wgs, lds, st);
        case 8: return launch_screen_ks<8>(P, cfg, wgs, lds, st);
        default: return launch_screen_ks<16>(P, cfg, wgs, lds, st);
    }
}

static int launch_rescore(const RescoreParams& R, hipStream_t st) {
    const int total = R.n_splits * R.L;
    const int dq = (R.d + 3) & ~3;
    const size_t rlds = (size_t)4 * ((size_t)(total + rescore_band_cap(total)) * sizeof(uint64_t) + (size_t)dq * sizeof(float) + 16);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_rescore_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(knn_rescore_kernel, dim3((unsigned)((R.q_end - R.q_begin + 3) / 4)), dim3(256), rlds, st, R);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

static int knn_screen_impl(const float* q16, const float* Xq, int64_t ldq, const float* norms_q, int64_t nq, int64_t q_offset,
                           const float* y16, const float* Y, int64_t ldy, const float* norms_y, int64_t n_db, int d, int k,
                           int metric, int exclude_self, int tier, int predict_unsplit, const uint32_t* meta, float* out_d,
                           int32_t* out_i, int32_t* flags, int32_t* n_flagged, void* ws, int64_t ws_bytes,
                           const ClusterTables* ct, int64_t q_pos_begin, int64_t q_pos_end, void* stream, int pred_L = 0,
                           int pred_terms = 0) {
    if (!q16 || !Xq || !norms_q || !y16 || !Y || !norms_y || !meta || !out_d || !out_i || !flags || !n_flagged || !ws)
        return TDR_ERR_BAD_ARG;
    if (nq <= 0 || n_db <= 0 || d <= 0 || ldq < d || ldy < d) return TDR_ERR_BAD_ARG;
    if (metric != 0 && metric != 1) return TDR_ERR_UNSUPPORTED;
    const int ks = pick_ks(d);
    if (ks == 0) return TDR_ERR_UNSUPPORTED;
    if (k < 1 || (int64_t)k > n_db - (exclude_self ? 1 : 0)) return TDR_ERR_BAD_ARG;
    if (tier < 0 || tier > 2) return TDR_ERR_BAD_ARG;
    const ScreenCfg cfg0 = screen_cfg(ks, k, tier);
    if (cfg0.L == 0) return TDR_ERR_UNSUPPORTED;
    // exact pruned searches keep lazy buffers (the approximate search stays on the sorted lists its CPU restatement was pinned with);
    // so do the pilots (predict_unsplit): every database slice of a sorted-list launch fills and sorts its own lists from scratch --
    // ~400 insertions per (query, slice), 3.7 / 7 ms for 512 queries over 64 slices -- whereas buffers take the same candidates by
    // plain stores.  A pilot compacts once more at the end and writes lists of the sorted form's length (same workspace, same
    // prediction: flags are raised for the band population an unsliced launch with THAT list length could not hold)
    const bool lazy_pilot = !ct && predict_unsplit;
    const ScreenCfg cfg = ((ct && ct->max_visit == 0) || lazy_pilot) ? lazy_cfg(ks, k, cfg0) : cfg0;
    const int L = cfg.L;
    const int L_out = (cfg.lazy && lazy_pilot) ? cfg0.L : L;
    if (n_db > 0x7fffffffLL) return TDR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    ScreenParams P;
    P.qp = q16; P.yp = y16; P.meta = meta; P.nq = nq; P.q_offset = q_offset; P.n_db = n_db; P.k = k; P.L = L;
    P.exclude_self = exclude_self;
    P.n_db_tiles = (int)((n_db + TILE_ROWS - 1) / TILE_ROWS);
    ScreenCfg cfg_split = cfg;
    cfg_split.L = L_out;    // the rescoring kernel's LDS holds splits x L_out entries
    P.n_splits = ct ? 1 : screen_splits(nq, P.n_db_tiles, cfg_split);  // the pruned scan walks clusters, not database slices
    P.L_out = L_out;
    P.tiles_per_split = (P.n_db_tiles + P.n_splits - 1) / P.n_splits;
    P.dpad = ks * 16;
    P.terms = cfg.terms;
    P.cand = (uint64_t*)ws;
    P.n_clusters = ct ? ct->n_clusters : 0;
    P.batch0 = 0;
    P.tile_cluster = ct ? ct->tile_cluster : nullptr; P.clus_tile_begin = ct ? ct->clus_tile_begin : nullptr;
    P.clus_radius = ct ? ct->clus_radius : nullptr; P.clus_dist = ct ? ct->clus_dist : nullptr;
    P.clus_order = ct ? ct->clus_order : nullptr;
    P.max_visit = ct ? ct->max_visit : 0;
    P.tile_cdist = ct ? ct->tile_cdist : nullptr;
    const int64_t lists_bytes = (int64_t)P.n_splits * nq * L_out * (int64_t)sizeof(uint64_t);
    const int64_t need = lists_bytes + (cfg.lazy ? ((nq * 4 + 15) / 16) * 16 : 0);
    if (ws_bytes < need) return TDR_ERR_WORKSPACE;
    P.lost = cfg.lazy ? (int32_t*)((char*)ws + lists_bytes) : nullptr;
    if (cfg.lazy && hipMemsetAsync(P.lost, 0, (size_t)nq * 4, st) != hipSuccess) return (int)hipGetLastError();
    int wgs = (int)((nq + 128 * cfg.qb - 1) / (128 * cfg.qb));
    int64_t q_lo = 0, q_hi = nq;
    if (ct && q_pos_end > q_pos_begin) {  // only the query batches covering [q_pos_begin, q_pos_end) of the sorted order
        const int64_t qpw = 128 * cfg.qb;
        P.batch0 = (int)(q_pos_begin / qpw);
        wgs = (int)((q_pos_end + qpw - 1) / qpw) - P.batch0;
        q_lo = q_pos_begin; q_hi = q_pos_end < nq ? q_pos_end : nq;
    }
    int rc = launch_lists_scan(P, cfg, ks, wgs, st);
    if (rc != TDR_OK) return rc;

    RescoreParams R;
    R.cand = P.cand; R.Xq = Xq; R.Y = Y; R.norms_q = norms_q; R.norms_y = norms_y; R.meta = meta; R.nq = nq; R.ldq = ldq;
    R.ldy = ldy; R.d = d; R.dpad = P.dpad; R.k = k; R.L = L_out; R.n_splits = P.n_splits; R.metric = metric; R.terms = cfg.terms; R.predict_unsplit = predict_unsplit; R.row_map = ct ? ct->row_map : nullptr; R.q_begin = q_lo; R.q_end = q_hi; R.out_d = out_d;
    R.out_i = out_i; R.flags = flags; R.n_flagged = n_flagged;
    R.pred_L = pred_L > 0 ? pred_L : L_out; R.pred_terms = pred_terms; R.lost = P.lost; R.unsorted = cfg.lazy;
    return launch_rescore(R, st);
}

int tdr_knn_screen_f32(const float* q16, const float* Xq, int64_t ldq, const float* norms_q, int64_t nq, int64_t q_offset,
                       const float* y16, const float* Y, int64_t ldy, const float* norms_y, int64_t n_db, int d, int k,
                       int metric, int exclude_self, int tier, int predict_unsplit, const uint32_t* meta, float* out_d,
                       int32_t* out_i, int32_t* flags, int32_t* n_flagged, void* ws, int64_t ws_bytes, void* stream) {
    return knn_screen_impl(q16, Xq, ldq, norms_q, nq, q_offset, y16, Y, ldy, norms_y, n_db, d, k, metric, exclude_self, tier,
                           predict_unsplit, meta, out_d, out_i, flags, n_flagged, ws, ws_bytes, nullptr, 0, 0, stream);
}

/* tdr_knn_screen_f32 with the pilot's prediction made for lists of pred_L entries (predict_unsplit = 1: flag the queries whose
 * error band holds >= pred_L candidates over all database slices): the threshold scan below keeps longer lists than the
 * list-keeping kernel can hold in LDS, and its tier is chosen by this prediction.  pred_terms = 0: the band of the pilot's own
 * tier; 2 (with tier >= 1): the band of the two-term split, counted on the three-term pilot's screening values. */
int tdr_knn_screen_pilot_f32(const float* q16, const float* Xq, int64_t ldq, const float* norms_q, int64_t nq, int64_t q_offset,
                             const float* y16, const float* Y, int64_t ldy, const float* norms_y, int64_t n_db, int d, int k,
                             int metric, int exclude_self, int tier, int pred_L, int pred_terms, const uint32_t* meta, float* out_d,
                             int32_t* out_i, int32_t* flags, int32_t* n_flagged, void* ws, int64_t ws_bytes, void* stream) {
    if (pred_terms != 0 && (pred_terms != 2 || tier < 1)) return TDR_ERR_BAD_ARG;
    return knn_screen_impl(q16, Xq, ldq, norms_q, nq, q_offset, y16, Y, ldy, norms_y, n_db, d, k, metric, exclude_self, tier,
                           1, meta, out_d, out_i, flags, n_flagged, ws, ws_bytes, nullptr, 0, 0, stream, pred_L, pred_terms);
}

/* ---- the UNPRUNED two-stage search as a threshold scan (csrc/tdr_knn_flat.hip) ------------------------------------------ */
int tdr_knn_flat_supported(int d);
int tdr_knn_flat_scan_f32(const float* q16, int64_t nq, int64_t q_offset, const float* y16, int64_t n_db, int d, int terms,
                          int exclude_self, int tile_begin, int tile_end, int tile_stride, const uint32_t* meta, const float* tau,
                          uint64_t* buf, int32_t* cnt, int cap, void* stream);
int tdr_knn_flat_select_f32(uint64_t* list, int have_list, const uint64_t* extra, const int32_t* extra_cnt, int n_sets,
                            int stride, const float* norms_q, const uint32_t* meta, int64_t nq, int d, int k, int L, int terms,
                            float* tau, int32_t* lost, void* stream);
int tdr_knn_flat_seed_f32(const float* q16, int64_t nq, int64_t q_offset, const float* y16, int64_t n_db, int d, int terms,
                          int exclude_self, int seed_tiles, int tile_stride, const uint32_t* meta, uint64_t* buf, int cap, void* stream);

namespace {
constexpr int FLAT_CAP = 256;        // appended entries a query may collect per pass
constexpr int FLAT_MIN_TILES = 4096; // database tiles below which the list-keeping kernel serves the search

constexpr int FLAT_MAX_PASSES = 30;
constexpr int FLAT_SEED_TILES = 8;   // 256 rows seed the lists (every screening value kept: FLAT_CAP entries per query)

struct FlatPlan {
    int ks, L, n_tiles, stride;
    int n_bounds, bounds[FLAT_MAX_PASSES + 2];   // pass i scans tile positions [bounds[i], bounds[i + 1])
    int64_t off_buf, off_cnt, off_tau, off_lost, total;   // byte offsets into the workspace (the lists sit at offset 0)
};

// seed = every row of the first 8 tile positions, then passes over ranges of positions growing geometrically up to the whole
// database, a select after each.  A pass that takes a query from n seen rows to r n appends ~ (r - 1) (k + B n / N) entries (B =
// the candidates inside the query's error band over the whole database, <= L - k or the query is flagged anyway), most in the last
// pass: (r - 1) (k + (L - k) / r).  The growth factor is the largest <= 4 that keeps this (+ 4 sigma) inside FLAT_CAP (k = 30: 4, six
// passes at N = 1M; k = 100: 2.55, nine), and the passes share the range evenly (ratio = (n_tiles / 8)^(1 / passes)): a pass that grew by 7.6 at
// N = 500k (r05_knn_flat_matrix.jsonl, first form of these bounds) lost 8 % of the queries to full buffers.
static bool flat_plan(int64_t nq, int64_t n_db, int d, int k, int terms, int L, FlatPlan* F) {
    F->ks = pick_ks(d);
    // 128 < d <= 256: the scan holds ONE query tile per wavefront and serves the one-term tier only (tdr_knn_flat.hip, flat_scan_ks)
    if (F->ks == 0 || F->ks > 16 || (F->ks == 16 && terms != 1) || terms < 1 || terms > 3 || L < k || L > 128 ||
        k > FLAT_SEED_TILES * 32 - 64)
        return false;
    F->n_tiles = (int)((n_db + TILE_ROWS - 1) / TILE_ROWS);
    if (F->n_tiles < FLAT_MIN_TILES) return false;
    F->L = L;
    // entries of the last pass: mean (r - 1) (k + (L - k) / r); the k-part is the number of later rows below the k-th order
    // statistic of the seen ones (variance = mean x r), the band part about Poisson -- mean + 4 sigma must fit the region
    double r = 4.0;
    for (; r > 1.5; r -= 0.05) {
        const double band = (r - 1.0) * (double)(L - k) / r, mean = (r - 1.0) * k + band;
        if (mean + 4.0 * sqrt((r - 1.0) * r * k + band) <= (double)FLAT_CAP) break;
    }
    const double span = (double)F->n_tiles / FLAT_SEED_TILES;
    int passes = (int)ceil(log(span) / log(r) - 1e-9);
    if (passes < 1) passes = 1;
    if (passes > FLAT_MAX_PASSES) return false;
    const double ratio = pow(span, 1.0 / passes);
    int nb = 0;
    F->bounds[nb++] = FLAT_SEED_TILES;
    double b = FLAT_SEED_TILES;
    for (int i = 1; i < passes; ++i) {
        b *= ratio;
        int e = ((int)(b + 0.5) + 1) & ~1;
        if (e <= F->bounds[nb - 1]) e = F->bounds[nb - 1] + 2;
        if (e >= F->n_tiles) break;
        F->bounds[nb++] = e;
    }
    F->bounds[nb++] = F->n_tiles;
    F->n_bounds = nb;
    // visiting order of the tiles: position j -> tile (j * stride) mod n_tiles, stride ~ 0.618 n_tiles and coprime to it: the
    // seed and every pass see rows from all over the database (a block sorted by class would otherwise take its thresholds
    // from one class and flood the buffers of every other)
    int st = (int)((double)F->n_tiles * 0.6180339887) | 1;
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    while (st < F->n_tiles && gcd(st, F->n_tiles) != 1) st += 2;
    F->stride = st < F->n_tiles ? st : 1;
    int64_t o = 0;
    o += nq * (int64_t)L * 8;                                        // lists (offset 0)
    F->off_buf = o;  o += nq * (int64_t)FLAT_CAP * 8;
    F->off_cnt = o;  o += ((nq * 4 + 15) / 16) * 16;
    F->off_tau = o;  o += ((nq * 4 + 15) / 16) * 16;
    F->off_lost = o; o += ((nq * 4 + 15) / 16) * 16;
    F->total = o;
    return true;
}
}  // namespace

/* Workspace bytes of tdr_knn_screen_flat_f32, 0 when the threshold scan does not serve the search (D > 256, D > 128 with more
 * than one term, a small database, L outside [k, 128], terms outside 1 .. 3). */
int64_t tdr_knn_screen_flat_workspace_bytes(int64_t nq, int64_t n_db, int d, int k, int terms, int L) {
    FlatPlan F;
    if (nq <= 0 || n_db <= 0 || k < 1 || !flat_plan(nq, n_db, d, k, terms, L, &F)) return 0;
    return F.total;
}

/* The pass plan of tdr_knn_screen_flat_f32 (host arithmetic only: no device is touched): bounds[0 .. n) = tile positions -- the seed
 * covers [0, bounds[0]), pass i scans [bounds[i], bounds[i + 1]), bounds[n - 1] = number of tiles; *stride = the visiting order's
 * stride (position j -> tile (j * stride) mod tiles).  Returns n (<= max_bounds), 0 when the threshold scan does not serve the
 * search, a negative error code for bad arguments. */
int tdr_knn_screen_flat_plan(int64_t nq, int64_t n_db, int d, int k, int terms, int L, int32_t* bounds, int max_bounds, int32_t* stride) {
    if (!bounds || max_bounds < 2 || nq <= 0 || n_db <= 0 || k < 1) return TDR_ERR_BAD_ARG;
    FlatPlan F;
    if (!flat_plan(nq, n_db, d, k, terms, L, &F)) return 0;
    if (F.n_bounds > max_bounds) return TDR_ERR_WORKSPACE;
    for (int i = 0; i < F.n_bounds; ++i) bounds[i] = F.bounds[i];
    if (stride) *stride = F.stride;
    return F.n_bounds;
}

/*
 * tdr_knn_screen_f32's contract (same operands, same outputs, same flags: flagged rows must be recomputed with
 * tdr_knn_packed_f32) for an UNPRUNED search of a large database, as seed -> threshold passes with selects -> rescoring
 * (csrc/tdr_knn_flat.hip).  terms = 1 (h.h'), 2 (h.h' + h.l': a third less matrix work than 3, half of 1's band) or 3; L = list length kept per query (k <= L <= 128; the band may hold L - k
 * candidates before a query is flagged).  Everything is enqueued on `stream`; nothing is read back.
 */
int tdr_knn_screen_flat_f32(const float* q16, const float* Xq, int64_t ldq, const float* norms_q, int64_t nq, int64_t q_offset,
                            const float* y16, const float* Y, int64_t ldy, const float* norms_y, int64_t n_db, int d, int k,
                            int metric, int exclude_self, int terms, int L, const uint32_t* meta, float* out_d, int32_t* out_i,
                            int32_t* flags, int32_t* n_flagged, void* ws, int64_t ws_bytes, void* stream) {
    if (!q16 || !Xq || !norms_q || !y16 || !Y || !norms_y || !meta || !out_d || !out_i || !flags || !n_flagged || !ws)
        return TDR_ERR_BAD_ARG;
    if (nq <= 0 || n_db <= 0 || d <= 0 || ldq < d || ldy < d) return TDR_ERR_BAD_ARG;
    if (metric != 0 && metric != 1) return TDR_ERR_UNSUPPORTED;
    if (k < 1 || (int64_t)k > n_db - (exclude_self ? 1 : 0) || n_db > 0x7fffffffLL) return TDR_ERR_BAD_ARG;
    FlatPlan F;
    if (!flat_plan(nq, n_db, d, k, terms, L, &F)) return TDR_ERR_UNSUPPORTED;
    if (ws_bytes < F.total) return TDR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    uint64_t* list = (uint64_t*)w;
    uint64_t* buf = (uint64_t*)(w + F.off_buf);
    int32_t* cnt = (int32_t*)(w + F.off_cnt);
    float* tau = (float*)(w + F.off_tau);
    int32_t* lost = (int32_t*)(w + F.off_lost);
    if (hipMemsetAsync(lost, 0, (size_t)nq * 4, st) != hipSuccess) return (int)hipGetLastError();

    // 1. seed: every screening value of the first 256 rows of the visiting order; 2. the first lists and thresholds
    int rc = tdr_knn_flat_seed_f32(q16, nq, q_offset, y16, n_db, d, terms, exclude_self, FLAT_SEED_TILES, F.stride, meta, buf, FLAT_CAP, stream);
    if (rc != TDR_OK) return rc;
    rc = tdr_knn_flat_select_f32(list, 0, buf, nullptr, 1, FLAT_CAP, norms_q, meta, nq, d, k, L, terms, tau, lost, stream);
    if (rc != TDR_OK) return rc;
    // 3. threshold passes over growing ranges of positions, a select after each
    for (int i = 0; i + 1 < F.n_bounds; ++i) {
        rc = tdr_knn_flat_scan_f32(q16, nq, q_offset, y16, n_db, d, terms, exclude_self, F.bounds[i], F.bounds[i + 1], F.stride, meta, tau, buf,
                                   cnt, FLAT_CAP, stream);
        if (rc != TDR_OK) return rc;
        rc = tdr_knn_flat_select_f32(list, 1, buf, cnt, 1, FLAT_CAP, norms_q, meta, nq, d, k, L, terms, tau, lost, stream);
        if (rc != TDR_OK) return rc;
    }
    // 4. rescoring of the final lists (one "split" of L entries per query)
    RescoreParams R;
    R.cand = list; R.Xq = Xq; R.Y = Y; R.norms_q = norms_q; R.norms_y = norms_y; R.meta = meta; R.nq = nq; R.ldq = ldq; R.ldy = ldy;
    R.d = d; R.dpad = F.ks * 16; R.k = k; R.L = L; R.n_splits = 1; R.metric = metric; R.terms = terms; R.predict_unsplit = 0;
    R.pred_L = L; R.pred_terms = 0; R.lost = lost; R.unsorted = 0; R.row_map = nullptr; R.q_begin = 0; R.q_end = nq; R.out_d = out_d; R.out_i = out_i;
    R.flags = flags; R.n_flagged = n_flagged;
    return launch_rescore(R, st);
}

/* Workspace bytes of the cluster-pruned searches below (tdr_knn_screen_clustered_f32 / _tb_f32 / tdr_knn_ivf_f32) over n_img image
 * rows: the candidate buffers of the launch they will make (lazy buffers: up to 126 entries per query + a word per query) --
 * never less than tdr_knn_screen_workspace_bytes(n_img, n_img, d, k, tier); 0 when the search is not supported. */
int64_t tdr_knn_screen_clustered_workspace_bytes(int64_t n_img, int d, int k, int tier) {
    const int ks = pick_ks(d);
    if (ks == 0 || n_img <= 0 || k < 1 || tier < 0 || tier > 2) return 0;
    const ScreenCfg c0 = screen_cfg(ks, k, tier);
    if (c0.L == 0) return 0;
    const ScreenCfg c = lazy_cfg(ks, k, c0);
    const int64_t plain = n_img * (int64_t)c0.L * 8;
    const int64_t lazy = c.lazy ? n_img * (int64_t)c.L * 8 + ((n_img * 4 + 15) / 16) * 16 : 0;
    return plain > lazy ? plain : lazy;
}

/* list form of the exact cluster-pruned searches and of the pilots: 1 (default since round 6) = unsorted candidate buffers,
 * compacted when full (screen_append_lazy in csrc/tdr_knn_screen.hip), 0 = the sorted lists of rounds 2-5; any other value
 * only reads the setting.  The same results either way (only which rows get flagged for the exact kernel can differ);
 * returns the previous value.  The host picks per search (distance/base.py:_pruned_launch: lazy where a query's work is its
 * own cluster, sorted where most of it is the steady scan of many clusters). */
int tdr_knn_screen_clustered_lists(int lazy) {
    const int old = g_clustered_lazy;
    if (lazy == 0 || lazy == 1) g_clustered_lazy = lazy;
    return old;
}

/*
 * Self search with cluster-bound pruning.  x16: fp16-split image of the n_img cluster-sorted, tile-padded rows
 * (tdr_pack16_mapped_f32 with row_map); X / norms: the n source rows.  row_map (n_img): image row -> source row or -1;
 * tile_cluster (n_img / 32), clus_tile_begin (n_clusters + 1), clus_radius (n_clusters, rounded up), clus_dist
 * (n_clusters^2 centre distances, rounded down), clus_order (n_clusters^2, clusters by increasing centre distance, self
 * first).  out_d / out_i / flags are indexed by SOURCE row and hold source indices; results are those of
 * tdr_knn_packed_f32 on the n source rows, whatever the clustering.  ws >= tdr_knn_screen_workspace_bytes(n_img, ...).
 * [q_pos_begin, q_pos_end): positions of the sorted order whose queries this launch answers (0, 0 = all; begin a multiple
 * of 256) -- a rank of a row-sharded search takes one contiguous range and the owners exchange the rows afterwards.
 */
int tdr_knn_screen_clustered_f32(const float* x16, const float* X, int64_t ldx, const float* norms, int64_t n_img, int d, int k,
                                 int metric, int exclude_self, int tier, const uint32_t* meta, const int32_t* row_map,
                                 int n_clusters, const int32_t* tile_cluster, const int32_t* clus_tile_begin,
                                 const float* clus_radius, const float* clus_dist, const int32_t* clus_order,
                                 int64_t q_pos_begin, int64_t q_pos_end, float* out_d, int32_t* out_i, int32_t* flags,
                                 int32_t* n_flagged, void* ws, int64_t ws_bytes, void* stream) {
    if (n_clusters > 4096) return TDR_ERR_UNSUPPORTED;     // the scanned-cluster bitmap of a workgroup (LDS) and the index builder's limit
    if (!row_map || !tile_cluster || !clus_tile_begin || !clus_radius || !clus_dist || !clus_order || n_clusters <= 0)
        return TDR_ERR_BAD_ARG;
    if (n_img % TILE_ROWS != 0) return TDR_ERR_BAD_ARG;
    if (q_pos_begin < 0 || q_pos_end > n_img || (q_pos_end > q_pos_begin && q_pos_begin % 256 != 0)) return TDR_ERR_BAD_ARG;
    ClusterTables ct = {n_clusters, row_map, tile_cluster, clus_tile_begin, clus_radius, clus_dist, clus_order, 0, nullptr};
    return knn_screen_impl(x16, X, ldx, norms, n_img, 0, x16, X, ldx, norms, n_img, d, k, metric, exclude_self, tier, 0, meta,
                           out_d, out_i, flags, n_flagged, ws, ws_bytes, &ct, q_pos_begin, q_pos_end, stream);
}

/* The same search with a per-tile table: tile_cdist (n_img / 32, n_clusters) holds, for every 32-row tile of the sorted
 * order, a LOWER bound of the distance (not squared) from any of its rows to every cluster centre
 * (tdr_cluster_tile_cdist_f32).  A cluster is then skipped when no row of the workgroup's query tiles can have a member
 * of it inside its band by |x - y| >= |x - c| - R_c -- sharper than the ball-to-ball bound (no query-side radius, and in high
 * dimension |x - c| ~ sqrt(|c_w - c|^2 + |x - c_w|^2)): clusters whose balls overlap are still told apart.  Results are the
 * same rows bit for bit; only the amount of skipped work differs. */
int tdr_knn_screen_clustered_tb_f32(const float* x16, const float* X, int64_t ldx, const float* norms, int64_t n_img, int d, int k,
                                    int metric, int exclude_self, int tier, const uint32_t* meta, const int32_t* row_map,
                                    int n_clusters, const int32_t* tile_cluster, const int32_t* clus_tile_begin,
                                    const float* clus_radius, const float* clus_dist, const int32_t* clus_order,
                                    const float* tile_cdist, int64_t q_pos_begin, int64_t q_pos_end, float* out_d, int32_t* out_i,
                                    int32_t* flags, int32_t* n_flagged, void* ws, int64_t ws_bytes, void* stream) {
    if (n_clusters > 4096) return TDR_ERR_UNSUPPORTED;
    if (!row_map || !tile_cluster || !clus_tile_begin || !clus_radius || !clus_dist || !clus_order || n_clusters <= 0)
        return TDR_ERR_BAD_ARG;
    if (n_img % TILE_ROWS != 0) return TDR_ERR_BAD_ARG;
    if (q_pos_begin < 0 || q_pos_end > n_img || (q_pos_end > q_pos_begin && q_pos_begin % 256 != 0)) return TDR_ERR_BAD_ARG;
    ClusterTables ct = {n_clusters, row_map, tile_cluster, clus_tile_begin, clus_radius, clus_dist, clus_order, 0, tile_cdist};
    return knn_screen_impl(x16, X, ldx, norms, n_img, 0, x16, X, ldx, norms, n_img, d, k, metric, exclude_self, tier, 0, meta,
                           out_d, out_i, flags, n_flagged, ws, ws_bytes, &ct, q_pos_begin, q_pos_end, stream);
}

/* Approximate (IVF-style) self search on the same cluster index: distance/faiss.py:331-349 (`IndexIVFFlat`, nlist =
 * n_clusters, nprobe).  Same arguments as tdr_knn_screen_clustered_f32; a workgroup (128 consecutive rows of the sorted
 * order) scans its wavefronts' own clusters (together they are probe 1) and then the nprobe - 1 clusters whose centres are
 * nearest to ANY of those, by (centre distance, id) -- a function of the index alone; a list the exact bound excludes keeps its
 * probe and is merely not scanned, which cannot change the result.  Candidates are rescored exactly, so
 * every returned distance is the reference's value for that pair; neighbours that live in unvisited clusters are missed.
 * Rows with fewer than k candidates leave the tail of out_d / out_i as the caller initialised it (+inf / -1). */
int tdr_knn_ivf_f32(const float* x16, const float* X, int64_t ldx, const float* norms, int64_t n_img, int d, int k, int metric,
                    int exclude_self, int tier, const uint32_t* meta, const int32_t* row_map, int n_clusters,
                    const int32_t* tile_cluster, const int32_t* clus_tile_begin, const float* clus_radius, const float* clus_dist,
                    const int32_t* clus_order, int nprobe, float* out_d, int32_t* out_i, int32_t* flags, int32_t* n_flagged,
                    void* ws, int64_t ws_bytes, void* stream) {
    if (n_clusters > 4096) return TDR_ERR_UNSUPPORTED;
    if (!row_map || !tile_cluster || !clus_tile_begin || !clus_radius || !clus_dist || !clus_order || n_clusters <= 0 || nprobe < 1)
        return TDR_ERR_BAD_ARG;
    if (n_img % TILE_ROWS != 0) return TDR_ERR_BAD_ARG;
    ClusterTables ct = {n_clusters, row_map, tile_cluster, clus_tile_begin, clus_radius, clus_dist, clus_order, nprobe, nullptr};
    return knn_screen_impl(x16, X, ldx, norms, n_img, 0, x16, X, ldx, norms, n_img, d, k, metric, exclude_self, tier, 0, meta,
                           out_d, out_i, flags, n_flagged, ws, ws_bytes, &ct, 0, 0, stream);
}

#ifdef TDR_SCREEN_STATS
/* measurement build only: copy the counters of the list-keeping scan to the host and (reset != 0) clear them */
int tdr_debug_screen_stats(unsigned long long* out, int reset) {
    static unsigned long long h[8][64];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(tdr::scr::g_screen_stats), sizeof(h)) != hipSuccess) return (int)hipGetLastError();
    for (int i = 0; i < 8; ++i) {
        out[i] = 0;
        for (int j = 0; j < 64; ++j) out[i] += h[i][j];
    }
    if (reset) {
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 64; ++j) h[i][j] = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(tdr::scr::g_screen_stats), h, sizeof(h)) != hipSuccess) return (int)hipGetLastError();
    }
    return TDR_OK;
}
#endif

}  // extern "C"
