// K1f -- the UNPRUNED screening scan of the two-stage exact kNN as a THRESHOLD scan (round 5).
//
// tdr_knn_screen.hip keeps, per query, a sorted list of the L smallest screening values in LDS and inserts into it while
// it scans: 48-63 KiB of lists per workgroup (so 128 queries per workgroup, a barrier per 32 database rows, 8 matrix
// instructions per barrier and wavefront in the one-term tier), ~450 cycles per insertion, and a matrix pipe that is busy
// 53 % of the time (profiles/r01_knn_screen_pmc.json).  That form is right for the cluster-pruned scan, whose thresholds must
// tighten WHILE it decides what to skip.  A scan that visits every tile anyway does not need lists:
//
//   seed      every screening value of 256 rows per query (knn_flat_seed_kernel): the k-th smallest screening value a query has met
//             is an upper bound of its k-th smallest over the whole database;
//   scan      (this file) the rest of the database is scanned against a FIXED per-query threshold tau = a_(k) + 2E: a
//             candidate is one fma + min tree + compare, and a survivor (a <= tau) is appended to the query's region in HBM -- no
//             lists, no insertion, nothing in LDS but the staged tiles;
//   select    one wavefront per query merges (list so far + appended) into the L smallest, sorted, and re-derives tau; the scan
//             runs in passes over ranges of tile positions that grow geometrically from the seed to the whole database (x4 for
//             k <= 30: six passes at N = 1M; the factor follows k so that a pass's ~ (r - 1) k entries fit the region:
//             flat_plan in tdr_knn_screen.hip) with a select after each;
//   rescore   the unchanged knn_rescore_kernel of tdr_knn_screen.hip on the final lists.
//
// Exactness is the list kernel's argument verbatim (tdr_knn_screen.hip header): tau_q >= a_(k)(whole database) + 2E at every
// moment, so every true neighbour passes; what a select drops lies beyond the L smallest seen so far and can never return;
// the rescoring kernel flags a query whose final list is full inside its band, and the host recomputes it exactly.
//
// What the freed LDS and registers buy: a wavefront holds TWO query tiles (QB = 2: 64 queries, the database fragments are
// read from LDS once for both), a step stages TWO database tiles (TPB = 2), so a barrier interval carries 4 blocks of 8
// (one term) .. 16 (two terms) matrix instructions instead of one; the finished block's 16 fma + min tree per lane runs in the
// shadow of the next block's matrix instructions (two accumulator sets, roles fixed at compile time: no register moves);
// with no list length to fit, the ONE-term tier (h.h' only, a third of the matrix work) serves data whose band holds up to
// ~100 candidates (lists of 128 live in HBM), the two-term tier (h.h' + h.l', half of that band, one query tile per wavefront)
// what it cannot, the three-term tier the rest.  128 < d <= 256: one term, one query tile per wavefront (flat_scan_ks).
#include "tdr_common.h"
#include "tdr_knn_screen_common.h"

namespace tdr {
namespace flat {

using scr::f16x8;
using scr::gptr_t;
using scr::lptr_t;
using scr::KEY_SENTINEL;

constexpr int NW = 4;        // wavefronts per workgroup
constexpr int WBUF = 128;    // survivor entries a wavefront buffers in LDS between flushes

struct FlatParams {
    const float* qp;       // fp16-split query images
    const float* yp;       // fp16-split database images
    const uint32_t* meta;
    int64_t nq, q_offset, n_db;
    int exclude_self;
    int t_begin, t_end;    // positions [t_begin, t_end) of the tile visiting order
    int n_tiles, stride;   // position j visits tile (j * stride) mod n_tiles (stride coprime to n_tiles; 1 = natural order): every
                           // range of positions is spread over the whole database, whatever order its rows come in
    int dpad;
    const float* tau;      // (nq) thresholds in screening units (a = c' + ||x||^2); +inf passes everything
    uint64_t* buf;         // (nq, cap) appended keys (screening value bits << 32 | database row)
    int32_t* cnt;          // (nq) candidates this launch met (> cap: the surplus was dropped)
    int cap;
};

__device__ __forceinline__ float reduce_tau(float tau, float xn) {
    return (tau - xn) + 2.3841858e-07f * (fabsf(tau) + xn);  // + 4u (|tau| + xn): never rejects an a <= tau
}

// TERMS = 1: h.h'   TERMS = 2: h.h' + h.l' (query h only)   TERMS = 3: h.h' + h.l' + l.h'
// A operand = database fragment (rows of the tile), B operand = query fragment; acc[r] of lane (q + 32 h) belongs to
// database row 8 (r >> 2) + 4 h + (r & 3) of the tile and query q of the query tile.
// Survivors: when a finished block holds any, its groups of four columns are tested and the hit columns' survivors go to a
// 128-entry buffer of the wavefront in LDS, emptied (to the queries' regions in HBM) at the start of a step or of a block's walk; a
// column the buffer cannot take (right after the seed, where per cents of all candidates survive; rows sorted by class, where the 64
// queries of a wavefront share their neighbours and meet a tile full of them) is appended straight to the regions.  A second,
// "dense" form of the walk (every column walked, flushes inside the walk) served the first passes until it was measured against
// this one on them: 0.7-2.8 % slower in six of six searches (profiles/r05_knn_flat_variants.json) -- removed.  (Testing and
// appending per group of four columns right where the group is reduced: +11 % on the big pass -- four more wave-wide tests per
// block on the hot path.)
// Variants measured and dropped (profiles/r05_knn_flat_variants.json): all matrix instructions of a block back to back before
// the arithmetic; both query tiles of a database tile on alternating accumulators; a three-deep staging ring with counted
// vmcnt waits -- all within 4 % of this form: the scan runs at the rate the matrix pipe sustains at the clock the chip holds
// under it (59 % busy at 1.79 GHz, r05_knn_flat_scan_pmc.json), not at a scheduling limit.
template <int KS, int TERMS, int QB, int TPB>
__global__ __launch_bounds__(256, 2) void knn_flat_scan_kernel(const FlatParams P) {
    static_assert((TPB * QB) % 2 == 0, "blocks per step must be even (static accumulator roles)");
    constexpr int TILE_LDS = KS * 1024 * (TERMS == 1 ? 1 : 2);   // staged bytes of one tile
    constexpr int STEP_LDS = TPB * TILE_LDS;
    constexpr int TILE_F = KS * 512 + 64;                          // floats per tile image in HBM
    constexpr int NBLK = 2 * KS;                                   // 1-KiB blocks per tile image
    constexpr int NPIECE = (TERMS == 1) ? KS : NBLK;               // 1-KiB pieces staged per tile
    constexpr int NSLOT = 4 * TPB;                                 // norm ring slots
    // SEPARATE LDS objects: the compiler orders an LDS store / atomic behind every pending LDS-DMA it cannot prove disjoint
    // (s_waitcnt vmcnt(0) -- it would wait for the NEXT step's tiles in the middle of this one); distinct variables are disjoint
    __shared__ __attribute__((aligned(16))) char stage0[2 * STEP_LDS];
    __shared__ __attribute__((aligned(16))) float nring[NSLOT * 64];
    __shared__ uint64_t wkeys_all[NW * WBUF];
    __shared__ uint32_t wq_all[NW * WBUF];
    __shared__ int cnt_all[NW * QB * 32];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 31, h = lane >> 5;
    uint64_t* wkeys = wkeys_all + wave * WBUF;
    uint32_t* wq = wq_all + wave * WBUF;
    int* cntw = cnt_all + wave * QB * 32;

    const int64_t n_qtiles = (P.nq + 31) / 32;
    const int64_t qt0 = ((int64_t)blockIdx.x * NW + wave) * QB;
    const bool wave_active = qt0 < n_qtiles;

    const int se = scr::scale_exp(P.meta[0]);
    const float m2s = -2.0f * scr::pow2f(-2 * se);

    f16x8 bh[QB][KS], bl[TERMS == 3 ? QB : 1][TERMS == 3 ? KS : 1];
    float xn[QB], tau_r[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int64_t qt = qt0 + qb;
        const bool blk_active = qt < n_qtiles;
        xn[qb] = 0.f;
        if (blk_active) {
            const char* qimg = reinterpret_cast<const char*>(P.qp + (size_t)qt * TILE_F);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bh[qb][s] = *reinterpret_cast<const f16x8*>(qimg + (2 * s) * 1024 + lane * 16);
                if constexpr (TERMS == 3) bl[qb][s] = *reinterpret_cast<const f16x8*>(qimg + (2 * s + 1) * 1024 + lane * 16);
            }
            xn[qb] = reinterpret_cast<const float*>(qimg + KS * 2048)[q];
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bh[qb][s][e] = (_Float16)0.f;
                if constexpr (TERMS == 3) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bl[qb][s][e] = (_Float16)0.f;
                }
            }
        }
        // rows beyond nq and padding rows (+inf norm) are not queries
        const bool lane_valid = blk_active && (qt * 32 + q < P.nq) && (xn[qb] < __builtin_inff());
        if (!lane_valid) xn[qb] = 0.f;
        tau_r[qb] = lane_valid ? reduce_tau(P.tau[qt * 32 + q], xn[qb]) : -__builtin_inff();
    }
    for (int p = lane; p < QB * 32; p += 64) cntw[p] = 0;
    int wcount = 0;       // wave-uniform: entries in this wavefront's survivor buffer

    const int n_steps = (P.t_end - P.t_begin + TPB - 1) / TPB;
    // tiles of the next position to stage / to multiply: position j visits tile (j * stride) mod n_tiles, walked by additions
    // (a 64-bit modulo per tile is a ~100-instruction software division on this hardware: +12 % on the whole scan when it sat here)
    int Tst = (int)(((int64_t)P.t_begin * P.stride) % P.n_tiles), Tcp = Tst;
    auto advance = [&](int& T) { T += P.stride; if (T >= P.n_tiles) T -= P.n_tiles; };

    auto stage = [&](int s) {
        char* dstbase = stage0 + (s & 1) * STEP_LDS;
#pragma unroll
        for (int tt = 0; tt < TPB; ++tt) {
            const int j = P.t_begin + s * TPB + tt;
            const int slot = (s * TPB + tt) & (NSLOT - 1);
            const int T = Tst;
            advance(Tst);
            if (j < P.t_end) {
                const float* src = P.yp + (size_t)T * TILE_F;
#pragma unroll
                for (int p0 = 0; p0 < NPIECE; p0 += NW) {
                    const int p = p0 + wave;
                    const int sblk = (TERMS == 1) ? 2 * p : p;
                    if (p < NPIECE)
                        __builtin_amdgcn_global_load_lds((gptr_t)(src + sblk * 256 + lane * 4),
                                                         (lptr_t)(dstbase + tt * TILE_LDS + p * 1024), 16, 0, 0);
                }
                if (wave == ((tt + 1) & 3))
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + NBLK * 256 + lane), (lptr_t)(nring + slot * 64), 4, 0, 0);
            } else if (wave == ((tt + 1) & 3)) {
                nring[slot * 64 + lane] = __builtin_inff();   // a tile beyond the range: nothing of it can pass
            }
        }
    };

    // buffered survivors -> the queries' regions in HBM (slot = the query's running count)
    auto flush = [&]() {
        for (int i = lane; i < wcount; i += 64) {
            const uint64_t key = wkeys[i];
            const uint32_t ql = wq[i];
            const int slot = atomicAdd(&cntw[ql], 1);
            if (slot < P.cap) P.buf[((size_t)qt0 * 32 + ql) * (size_t)P.cap + slot] = key;
        }
        wcount = 0;   // (LDS operations of a wavefront complete in order: later buffer writes cannot overtake these reads)
    };

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    // norms of a block's tile rows are read while ITS matrix instructions run and used one block later, when it is finished
    // (yn[c] belongs to the block whose accumulators are acc[c]); the first block of the scan finishes a block of +inf norms
    f32x4 yn[2][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { yn[0][g] = f32x4{__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()}; yn[1][g] = yn[0][g]; }
    int Tprev = 0;
    const uint32_t n_db32 = (uint32_t)P.n_db;

    // one column of the finished block: append the lanes whose candidate survives
    auto take_column = [&](float v, float tq, float xq, uint32_t j, uint32_t jself, int pq) {
        const bool pass = v <= tq && v < __builtin_inff() && j < n_db32 && j != jself;
        const unsigned long long m = __ballot(pass);
        if (m == 0ull) return;
        const int nb = __popcll(m);
        if (__builtin_amdgcn_readfirstlane(wcount + nb) > WBUF) {
            // the buffer cannot take this column: its survivors go straight to the queries' regions, nothing is lost
            if (pass) {
                const int ql = pq * 32 + q;
                const int slot = atomicAdd(&cntw[ql], 1);
                if (slot < P.cap) P.buf[((size_t)qt0 * 32 + ql) * (size_t)P.cap + slot] = mkkey(v + xq, j);
            }
            return;
        }
        if (pass) {
            const int pos = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            wkeys[pos] = mkkey(v + xq, j);
            wq[pos] = (uint32_t)(pq * 32 + q);
        }
        wcount = __builtin_amdgcn_readfirstlane(wcount + nb);
    };
    auto query_of = [&](int pq, float& tq, float& xq, uint32_t& jself) {
        tq = tau_r[0]; xq = xn[0];
        int64_t js64 = (qt0 * 32 + q) + P.q_offset;
#pragma unroll
        for (int b = 1; b < QB; ++b)
            if (pq == b) { tq = tau_r[b]; xq = xn[b]; js64 = ((qt0 + b) * 32 + q) + P.q_offset; }
        // database rows are < 2^31; a query index beyond that range matches none of them
        jself = (P.exclude_self && js64 >= 0 && js64 < 0x7fffffffLL) ? (uint32_t)js64 : 0xffffffffu;
    };

    // group g of the finished block: c' = ||y||^2 - 2 s^-2 acc of its four columns, their minimum
    auto finish_part = [&](const f32x16& a, const f32x4 (&ynb)[4], int g, float (&dv)[16], float (&pm)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dv[4 * g + e] = __builtin_fmaf(m2s, a[4 * g + e], ynb[g][e]);
        pm[g] = fminf(fminf(dv[4 * g], dv[4 * g + 1]), fminf(dv[4 * g + 2], dv[4 * g + 3]));
    };
    // the finished block's survivors (the buffer is emptied here, at the head of the walk, when it is half full)
    auto survivors = [&](const float (&dv)[16], const float (&pm)[4], int pq) {
        float tq, xq;
        uint32_t jself;
        query_of(pq, tq, xq, jself);
        const uint32_t jb = (uint32_t)Tprev * 32u + 4u * (uint32_t)h;
        if (wcount >= WBUF / 2) flush();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (!__any(pm[g] <= tq)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) take_column(dv[4 * g + e], tq, xq, jb + (uint32_t)(e + 8 * g), jself, pq);
        }
    };

    // one block: the matrix instructions of (tile fragments A, query block qb) into acc[CUR]; the finished block acc[1 - CUR]
    // is reduced between them.  ROTATE: the fragments of the NEXT tile of this step replace each slice as soon as its last
    // matrix instruction has been issued.
#define TDR_FLAT_MMA(ACC, S, FIRST)                                                                                   \
    do {                                                                                                              \
        ACC = (FIRST) ? __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[S], bh[qb][S], zero16, 0, 0, 0)                     \
                      : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[S], bh[qb][S], ACC, 0, 0, 0);                       \
        if constexpr (TERMS >= 2) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[S], bh[qb][S], ACC, 0, 0, 0);       \
        if constexpr (TERMS == 3) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[S], bl[qb][S], ACC, 0, 0, 0);       \
    } while (0)

    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    f16x8 ah[KS], al[TERMS >= 2 ? KS : 1];

    auto load_frags = [&](const char* img) {
        const char* ap = img + lane * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if constexpr (TERMS == 1) ah[s] = *reinterpret_cast<const f16x8*>(ap + s * 1024);
            else {
                ah[s] = *reinterpret_cast<const f16x8*>(ap + (2 * s) * 1024);
                al[s] = *reinterpret_cast<const f16x8*>(ap + (2 * s + 1) * 1024);
            }
        }
    };

    stage(0);
    __syncthreads();
    for (int s = 0; s < n_steps; ++s) {
        // first the flush (its stores and the LDS atomics the compiler orders behind pending LDS-DMA meet an empty queue), then
        // the request for the next step's tiles
        if (wave_active && wcount >= WBUF / 2) flush();
        if (s + 1 < n_steps) stage(s + 1);
        if (wave_active) {
            const char* sbase = stage0 + (s & 1) * STEP_LDS;
#pragma unroll
            for (int tt = 0; tt < TPB; ++tt) {
                const int T = Tcp;
                advance(Tcp);
                if (tt == 0) load_frags(sbase);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const int cur = (tt * QB + qb) & 1;          // compile-time after unrolling
                    const int pq = (qb == 0) ? QB - 1 : qb - 1;  // query block of the finished block
                    const bool rotate = (qb == QB - 1) && (tt + 1 < TPB);
                    const char* nap = sbase + (tt + 1) * TILE_LDS + lane * 16;
                    float dv[16], pm[4];
                    {   // this block's own norms, for the moment it is finished (one block later)
                        const float* ynp = nring + ((s * TPB + tt) & (NSLOT - 1)) * 64 + 4 * h;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            if (cur == 0) yn[0][g] = *reinterpret_cast<const f32x4*>(ynp + 8 * g);
                            else yn[1][g] = *reinterpret_cast<const f32x4*>(ynp + 8 * g);
                        }
                    }
                    constexpr int PARTS = 4;
                    constexpr int SPP = (KS + PARTS - 1) / PARTS;   // slices per part
#pragma unroll
                    for (int part = 0; part < PARTS; ++part) {
#pragma unroll
                        for (int u = 0; u < SPP; ++u) {
                            const int s2 = part * SPP + u;
                            if (s2 < KS) {
                                if (cur == 0) TDR_FLAT_MMA(acc[0], s2, s2 == 0);
                                else TDR_FLAT_MMA(acc[1], s2, s2 == 0);
                                if (rotate) {
                                    if constexpr (TERMS == 1) ah[s2] = *reinterpret_cast<const f16x8*>(nap + s2 * 1024);
                                    else {
                                        ah[s2] = *reinterpret_cast<const f16x8*>(nap + (2 * s2) * 1024);
                                        al[s2] = *reinterpret_cast<const f16x8*>(nap + (2 * s2 + 1) * 1024);
                                    }
                                }
                            }
                        }
                        if (cur == 0) finish_part(acc[1], yn[1], part, dv, pm);
                        else finish_part(acc[0], yn[0], part, dv, pm);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    {
                        float tq = tau_r[0];
#pragma unroll
                        for (int b = 1; b < QB; ++b)
                            if (pq == b) tq = tau_r[b];
                        const float mn = fminf(fminf(pm[0], pm[1]), fminf(pm[2], pm[3]));
                        if (__any(mn <= tq)) survivors(dv, pm, pq);
                    }
                    Tprev = T;   // this block is the next one's finished block
                }
            }
        }
        __syncthreads();
    }
    if (wave_active) {
        // drain: the last block (its accumulator set: blocks per step is even, so it is acc[1])
        float dv[16], pm[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) finish_part(acc[1], yn[1], g, dv, pm);
        const float mn = fminf(fminf(pm[0], pm[1]), fminf(pm[2], pm[3]));
        if (__any(mn <= tau_r[QB - 1])) survivors(dv, pm, QB - 1);
        flush();
        for (int p = lane; p < QB * 32; p += 64) {
            const int64_t qi = qt0 * 32 + p;
            if (qi < P.nq) P.cnt[qi] = cntw[p];     // > cap: the entries beyond the region were dropped, the select marks the query lost
        }
    }
#undef TDR_FLAT_MMA
}

// ---------------------------------------------------------------------------------------------------------
// Seed: the screening values of EVERY row of the first `t_end` tile positions, per query, straight into the query's buffer
// (key (a, row); the query itself, rows beyond the database and padding rows as sentinels).  One wavefront per query tile,
// fragments read from HBM directly (a few tiles: nothing to stage).  The select that follows turns them into the first list
// and the first threshold; short DENSE passes over ranges growing by four then bring every query to the point (1/64 of the
// database seen) from which the three long passes start.  The list-keeping kernel as pilot over that 1/64 cost 33-35 ms at
// N = 1M whatever its list length (its sorted insertions are front-loaded); seed + three short passes + their selects ~22.
// ---------------------------------------------------------------------------------------------------------
template <int KS, int TERMS>
__global__ __launch_bounds__(256) void knn_flat_seed_kernel(const FlatParams P) {
    constexpr int TILE_F = KS * 512 + 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 31, h = lane >> 5;
    const int64_t n_qtiles = (P.nq + 31) / 32;
    const int64_t qt = (int64_t)blockIdx.x * NW + wave;
    if (qt >= n_qtiles) return;
    const int se = scr::scale_exp(P.meta[0]);
    const float m2s = -2.0f * scr::pow2f(-2 * se);
    const char* qimg = reinterpret_cast<const char*>(P.qp + (size_t)qt * TILE_F);
    f16x8 bh[KS], bl[TERMS == 3 ? KS : 1];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        bh[s] = *reinterpret_cast<const f16x8*>(qimg + (2 * s) * 1024 + lane * 16);
        if constexpr (TERMS == 3) bl[s] = *reinterpret_cast<const f16x8*>(qimg + (2 * s + 1) * 1024 + lane * 16);
    }
    float xq = reinterpret_cast<const float*>(qimg + KS * 2048)[q];
    const int64_t qi = qt * 32 + q;
    const bool q_valid = qi < P.nq && xq < __builtin_inff();
    if (!q_valid) xq = 0.f;
    const int64_t js64 = qi + P.q_offset;
    const uint32_t jself = (P.exclude_self && js64 >= 0 && js64 < 0x7fffffffLL) ? (uint32_t)js64 : 0xffffffffu;
    const uint32_t n_db32 = (uint32_t)P.n_db;
    int T = (int)(((int64_t)P.t_begin * P.stride) % P.n_tiles);
    for (int jt = P.t_begin; jt < P.t_end; ++jt) {
        const char* img = reinterpret_cast<const char*>(P.yp + (size_t)T * TILE_F);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(img + (2 * s) * 1024 + lane * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], acc, 0, 0, 0);
            if constexpr (TERMS >= 2) {
                const f16x8 al = *reinterpret_cast<const f16x8*>(img + (2 * s + 1) * 1024 + lane * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s], acc, 0, 0, 0);
            }
            if constexpr (TERMS == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s], acc, 0, 0, 0);
        }
        const float* ynp = reinterpret_cast<const float*>(img + KS * 2048) + 4 * h;
        if (qi < P.nq) {   // a query whose packed norm is +inf (non-finite or overflowing row) gets sentinels, not an unwritten region
            uint64_t* dst = P.buf + (size_t)qi * (size_t)P.cap + (size_t)(jt - P.t_begin) * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 y4 = *reinterpret_cast<const f32x4*>(ynp + 8 * g);
                uint64_t k4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t j = (uint32_t)T * 32u + (uint32_t)(8 * g + 4 * h + e);
                    const float c = __builtin_fmaf(m2s, acc[4 * g + e], y4[e]);
                    const bool ok = q_valid && c < __builtin_inff() && j < n_db32 && j != jself;
                    k4[e] = ok ? mkkey(c + xq, j) : KEY_SENTINEL;
                }
                // rows 8 g + 4 h .. + 3: four consecutive keys (32 bytes, sector-aligned)
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                u64x2* d2 = reinterpret_cast<u64x2*>(dst + 8 * g + 4 * h);
                d2[0] = u64x2{k4[0], k4[1]};
                d2[1] = u64x2{k4[2], k4[3]};
            }
        }
        T += P.stride;
        if (T >= P.n_tiles) T -= P.n_tiles;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Nearest centre of every point (cluster index of the pruned search, step 3): one-term screening values against the C
// centres, arg-min per point.  The assignment only shapes the clusters -- the search result never depends on it -- so the
// 2^-10 relative error of h.h' is irrelevant, and the kernel is deterministic (every rank builds the same index).  Replaces
// an exact fp32-MFMA search with k = 1 (3.8 - 6.4 ms at N = 1M, C = 1000) by ~0.4 ms on the critical path of the kNN build.
// ---------------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void nearest_centre_kernel(const float* __restrict__ x16, int64_t n, const float* __restrict__ c16,
                                                             int n_centres, const uint32_t* __restrict__ meta, int32_t* __restrict__ labels) {
    constexpr int TILE_F = KS * 512 + 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 31, h = lane >> 5;
    const int64_t n_qtiles = (n + 31) / 32;
    const int64_t qt = (int64_t)blockIdx.x * NW + wave;
    if (qt >= n_qtiles) return;
    const int se = scr::scale_exp(meta[0]);
    const float m2s = -2.0f * scr::pow2f(-2 * se);
    const char* qimg = reinterpret_cast<const char*>(x16 + (size_t)qt * TILE_F);
    f16x8 bh[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) bh[s] = *reinterpret_cast<const f16x8*>(qimg + (2 * s) * 1024 + lane * 16);
    float best = __builtin_inff();
    int arg = 0x7fffffff;
    const int c_tiles = (n_centres + 31) / 32;
    for (int T = 0; T < c_tiles; ++T) {
        const char* img = reinterpret_cast<const char*>(c16 + (size_t)T * TILE_F);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(img + (2 * s) * 1024 + lane * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], acc, 0, 0, 0);
        }
        const float* ynp = reinterpret_cast<const float*>(img + KS * 2048) + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(ynp + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = T * 32 + 8 * g + 4 * h + e;
                const float v = __builtin_fmaf(m2s, acc[4 * g + e], y4[e]);     // ||c||^2 - 2 x.c (+inf for padding rows)
                if (c < n_centres && (v < best || (v == best && c < arg))) { best = v; arg = c; }
            }
        }
    }
    // the two lanes of a point (rows 4 h .. of every group) meet
    const float ob = __shfl_xor(best, 32, 64);
    const int oa = __shfl_xor(arg, 32, 64);
    if (ob < best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    const int64_t qi = qt * 32 + q;
    if (h == 0 && qi < n) labels[qi] = arg == 0x7fffffff ? 0 : arg;
}

// ---------------------------------------------------------------------------------------------------------
// Select: one wavefront per query.  (sorted list of L keys so far, may be NULL) + (n_extra keys in any order: the scan's
// appended entries, or the pilot's per-slice lists) -> the L smallest, ascending, sentinel-padded; tau = min(a_(k) + 2E,
// a_(L) if the list is full) for the next pass.  A query whose appended count exceeded its capacity lost entries: its
// `lost` word is set (the host recomputes it exactly).
// ---------------------------------------------------------------------------------------------------------
struct SelectParams {
    uint64_t* list;          // (nq, L) in/out
    int have_list;           // 0: the list is empty on entry (first call)
    const uint64_t* extra;   // (n_sets, nq, stride) keys
    const int32_t* extra_cnt;// (nq) valid entries per query (NULL: all `stride`, sentinels allowed)
    int n_sets, stride;
    const float* norms_q;    // (nq) reference-order squared norms of the queries (screening order)
    const uint32_t* meta;
    int64_t nq;
    int k, L, dpad, terms;
    float* tau;              // (nq) out
    int32_t* lost;           // (nq) |= 1 where extra_cnt > stride
};

__global__ __launch_bounds__(256) void knn_flat_select_kernel(const SelectParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int maxE = P.n_sets * P.stride;
    uint64_t* lk = reinterpret_cast<uint64_t*>(smem_raw) + (size_t)wave * (P.L + maxE + P.L);  // list | extras | out
    uint64_t* ek = lk + P.L;
    uint64_t* ok = ek + maxE;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
    if (qi >= P.nq) return;   // wavefronts are independent (no block-level barrier below)
    int nE = 0;
    bool lost = false;
    if (P.extra_cnt) {
        const int c = P.extra_cnt[qi];
        lost = c > P.stride;               // the pass met more candidates than the region holds: the query is recomputed exactly
        nE = c > P.stride ? P.stride : (c < 0 ? 0 : c);
        for (int p = lane; p < nE; p += 64) ek[p] = P.extra[(size_t)qi * P.stride + p];
    } else {
        nE = maxE;
        for (int p = lane; p < maxE; p += 64) {
            const int s = p / P.stride, r = p - s * P.stride;
            ek[p] = P.extra[((size_t)s * P.nq + qi) * P.stride + r];
        }
    }
    const int nL = P.have_list ? P.L : 0;
    for (int p = lane; p < nL; p += 64) lk[p] = P.list[(size_t)qi * P.L + p];
    for (int p = lane; p < P.L; p += 64) ok[p] = KEY_SENTINEL;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // rank of every key among (list + extras): a list entry's rank = its position + extras below it; an extra's rank =
    // list entries below it + extras below it (keys are distinct except sentinels, which are never placed).  A lane keeps its
    // keys (list entries lane, lane + 64; extras lane + 64 t) in registers and walks the LDS copies once.
    constexpr int MAXE_PER_LANE = 32;   // maxE <= 2048
    uint64_t ml[2];
    int rl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = lane + 64 * t;
        ml[t] = (p < nL) ? lk[p] : KEY_SENTINEL;
        rl[t] = p;
    }
    for (int e0 = 0; e0 < nE; e0 += 64 * 4) {
        // four extras per lane and round
        uint64_t me[4];
        int re[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = e0 + lane + 64 * t;
            me[t] = (p < nE) ? ek[p] : KEY_SENTINEL;
            re[t] = 0;
        }
        for (int e = 0; e < nE; ++e) {
            const uint64_t ke = ek[e];
#pragma unroll
            for (int t = 0; t < 4; ++t) re[t] += (ke < me[t]) ? 1 : 0;
        }
        if (nL > 0) {
            // the list is ascending (sentinels last): the number of its entries below a key is a lower bound search, 7-8 probes
            // instead of a walk over all 128
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int lo = 0, hi = nL;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (lk[mid] < me[t]) lo = mid + 1;
                    else hi = mid;
                }
                re[t] += lo;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = e0 + lane + 64 * t;
            if (p < nE && me[t] != KEY_SENTINEL && re[t] < P.L) ok[re[t]] = me[t];
        }
    }
    (void)MAXE_PER_LANE;
    if (nL > 0) {
        for (int e = 0; e < nE; ++e) {
            const uint64_t ke = ek[e];
#pragma unroll
            for (int t = 0; t < 2; ++t) rl[t] += (ke < ml[t]) ? 1 : 0;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p = lane + 64 * t;
            if (p < nL && ml[t] != KEY_SENTINEL && rl[t] < P.L) ok[rl[t]] = ml[t];
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int p = lane; p < P.L; p += 64) P.list[(size_t)qi * P.L + p] = ok[p];
    if (lane == 0) {
        const uint64_t kk = ok[P.k - 1], kl = ok[P.L - 1];
        float tau = __builtin_inff();
        if (kk != KEY_SENTINEL) {
            const int se = scr::scale_exp(P.meta[0]);
            const float band = scr::screen_band(P.norms_q[qi], __uint_as_float(P.meta[1]), P.dpad, se, P.terms);
            tau = u2f((uint32_t)(kk >> 32)) + band;
            if (kl != KEY_SENTINEL) tau = fminf(tau, u2f((uint32_t)(kl >> 32)));
        }
        P.tau[qi] = tau;
        if (lost) P.lost[qi] = 1;
    }
}

static int flat_ks(int d) {
    if (d <= 32) return 2;
    if (d <= 64) return 4;
    if (d <= 128) return 8;
    return 0;
}
// the scan and its seed also serve 128 < d <= 256 with ONE term and one query tile per wavefront (64 registers of query
// fragments, 16-KiB tiles: 70 KiB of LDS per workgroup, two workgroups per CU)
static int flat_scan_ks(int d, int terms) {
    if (d <= 128) return flat_ks(d);
    return (d <= 256 && terms == 1) ? 16 : 0;
}

template <int KS, int TERMS, int QB, int TPB>
static int launch_flat(const FlatParams& P, hipStream_t st) {
    const int64_t n_qtiles = (P.nq + 31) / 32;
    const int64_t wgs = (n_qtiles + NW * QB - 1) / (NW * QB);
    hipLaunchKernelGGL((knn_flat_scan_kernel<KS, TERMS, QB, TPB>), dim3((unsigned)wgs), dim3(256), 0, st, P);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

// shape of the workgroup per tier: one term keeps two query tiles per wavefront (the query's h fragments are 64 VGPRs), two and
// three terms one; 128 < d <= 256 (KS = 16): one term, one query tile
template <int KS>
static int launch_flat_ks(const FlatParams& P, int terms, hipStream_t st) {
    if constexpr (KS == 16) {
        if (terms != 1) return TDR_ERR_UNSUPPORTED;
        return launch_flat<16, 1, 1, 2>(P, st);
    } else {
        if (terms == 1) return launch_flat<KS, 1, 2, 2>(P, st);
        if (terms == 2) return launch_flat<KS, 2, 1, 2>(P, st);
        return launch_flat<KS, 3, 1, 2>(P, st);
    }
}

}  // namespace flat
}  // namespace tdr

using namespace tdr;

extern "C" {

/* 1 when the f16 kernels of this file that hold two query tiles per wavefront serve feature dimension d (<= 128; the threshold
 * scan itself also serves d <= 256 with one term: tdr_knn_screen_flat_workspace_bytes says what it takes). */
int tdr_knn_flat_supported(int d) { return flat::flat_ks(d) != 0 ? 1 : 0; }
/* 1 when tdr_cluster_assign16_f32 serves feature dimension d (<= 256). */
int tdr_cluster_assign16_supported(int d) { return flat::flat_scan_ks(d, 1) != 0 ? 1 : 0; }

/*
 * One pass of the threshold scan (tdr_knn_flat.hip header): every candidate of the database tiles at positions [tile_begin,
 * tile_end) of the visiting order (position j = tile (j * tile_stride) mod n_tiles; tile_stride coprime to n_tiles, 1 = natural) whose
 * screening value is <= tau[q] is appended to buf[q * cap ...]; cnt[q] = the number met (entries beyond cap are dropped: the
 * caller treats cnt > cap as lost).  q16 / y16: fp16-split images packed with the same meta; terms = 1, 2 or 3 (see
 * screen_band in tdr_knn_screen_common.h for the band each needs).
 */
int tdr_knn_flat_scan_f32(const float* q16, int64_t nq, int64_t q_offset, const float* y16, int64_t n_db, int d, int terms,
                          int exclude_self, int tile_begin, int tile_end, int tile_stride, const uint32_t* meta, const float* tau,
                          uint64_t* buf, int32_t* cnt, int cap, void* stream) {
    if (!q16 || !y16 || !meta || !tau || !buf || !cnt || nq <= 0 || n_db <= 0 || d <= 0 || cap <= 0) return TDR_ERR_BAD_ARG;
    if (terms < 1 || terms > 3) return TDR_ERR_BAD_ARG;
    const int ks = flat::flat_scan_ks(d, terms);
    if (ks == 0) return TDR_ERR_UNSUPPORTED;
    const int n_tiles = (int)((n_db + scr::TILE_ROWS - 1) / scr::TILE_ROWS);
    if (tile_begin < 0 || tile_end > n_tiles || tile_begin >= tile_end || tile_stride < 1) return TDR_ERR_BAD_ARG;
    {   // the visiting order must be a permutation of the tiles: stride below the tile count (the kernel walks it by additions)
        // and coprime to it
        if (tile_stride >= n_tiles && n_tiles > 1) return TDR_ERR_BAD_ARG;
        int a = tile_stride, b = n_tiles;
        while (b) { const int t = a % b; a = b; b = t; }
        if (a != 1) return TDR_ERR_BAD_ARG;
    }
    if (n_db > 0x7fffffffLL) return TDR_ERR_UNSUPPORTED;
    flat::FlatParams P;
    P.qp = q16; P.yp = y16; P.meta = meta; P.nq = nq; P.q_offset = q_offset; P.n_db = n_db; P.exclude_self = exclude_self;
    P.t_begin = tile_begin; P.t_end = tile_end; P.dpad = ks * 16; P.tau = tau; P.buf = buf; P.cnt = cnt; P.cap = cap;
    P.n_tiles = n_tiles; P.stride = tile_stride;
    hipStream_t st = (hipStream_t)stream;
    switch (ks) {
        case 2: return flat::launch_flat_ks<2>(P, terms, st);
        case 4: return flat::launch_flat_ks<4>(P, terms, st);
        case 16: return flat::launch_flat_ks<16>(P, terms, st);
        default: return flat::launch_flat_ks<8>(P, terms, st);
    }
}

/*
 * Screening values of every row of the tiles at positions [0, seed_tiles) of the visiting order into buf[q * cap + 32 j + row]
 * (cap >= 32 seed_tiles, a multiple of 4; sentinels for the query itself, padding rows and rows beyond the database) -- the seed
 * of the threshold scan's lists.
 */
int tdr_knn_flat_seed_f32(const float* q16, int64_t nq, int64_t q_offset, const float* y16, int64_t n_db, int d, int terms,
                          int exclude_self, int seed_tiles, int tile_stride, const uint32_t* meta, uint64_t* buf, int cap, void* stream) {
    if (!q16 || !y16 || !meta || !buf || nq <= 0 || n_db <= 0 || d <= 0 || seed_tiles < 1) return TDR_ERR_BAD_ARG;
    if (terms < 1 || terms > 3 || cap < 32 * seed_tiles || (cap & 3) != 0) return TDR_ERR_BAD_ARG;
    const int ks = flat::flat_scan_ks(d, terms);
    if (ks == 0) return TDR_ERR_UNSUPPORTED;
    const int n_tiles = (int)((n_db + scr::TILE_ROWS - 1) / scr::TILE_ROWS);
    if (seed_tiles > n_tiles || n_db > 0x7fffffffLL || tile_stride < 1 || (tile_stride >= n_tiles && n_tiles > 1)) return TDR_ERR_BAD_ARG;
    flat::FlatParams P;
    P.qp = q16; P.yp = y16; P.meta = meta; P.nq = nq; P.q_offset = q_offset; P.n_db = n_db; P.exclude_self = exclude_self;
    P.t_begin = 0; P.t_end = seed_tiles; P.dpad = ks * 16; P.tau = nullptr; P.buf = buf; P.cnt = nullptr; P.cap = cap;
    P.n_tiles = n_tiles; P.stride = tile_stride;
    const int64_t n_qtiles = (nq + 31) / 32;
    const dim3 grid((unsigned)((n_qtiles + flat::NW - 1) / flat::NW));
    hipStream_t st = (hipStream_t)stream;
#define TDR_SEED(KS_)                                                                                                  \
    do {                                                                                                               \
        if (terms == 1) hipLaunchKernelGGL((flat::knn_flat_seed_kernel<KS_, 1>), grid, dim3(256), 0, st, P);           \
        else if (terms == 2) hipLaunchKernelGGL((flat::knn_flat_seed_kernel<KS_, 2>), grid, dim3(256), 0, st, P);      \
        else hipLaunchKernelGGL((flat::knn_flat_seed_kernel<KS_, 3>), grid, dim3(256), 0, st, P);                      \
    } while (0)
    switch (ks) {
        case 2: TDR_SEED(2); break;
        case 4: TDR_SEED(4); break;
        case 16: hipLaunchKernelGGL((flat::knn_flat_seed_kernel<16, 1>), grid, dim3(256), 0, st, P); break;
        default: TDR_SEED(8); break;
    }
#undef TDR_SEED
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* labels[i] = the centre nearest to point i by the one-term screening value (x16 / c16: fp16-split images of the n points and
 * of the n_centres centres, packed with the same meta).  An approximate arg-min (2^-10 relative): for building clusters. */
int tdr_cluster_assign16_f32(const float* x16, int64_t n, const float* c16, int n_centres, int d, const uint32_t* meta,
                             int32_t* labels, void* stream) {
    if (!x16 || !c16 || !meta || !labels || n <= 0 || n_centres <= 0 || d <= 0) return TDR_ERR_BAD_ARG;
    const int ks = flat::flat_scan_ks(d, 1);     // one term: 128 < d <= 256 too (64 registers of query fragments)
    if (ks == 0) return TDR_ERR_UNSUPPORTED;
    const int64_t n_qtiles = (n + 31) / 32;
    const dim3 grid((unsigned)((n_qtiles + flat::NW - 1) / flat::NW));
    hipStream_t st = (hipStream_t)stream;
    switch (ks) {
        case 2: hipLaunchKernelGGL(flat::nearest_centre_kernel<2>, grid, dim3(256), 0, st, x16, n, c16, n_centres, meta, labels); break;
        case 4: hipLaunchKernelGGL(flat::nearest_centre_kernel<4>, grid, dim3(256), 0, st, x16, n, c16, n_centres, meta, labels); break;
        case 16: hipLaunchKernelGGL(flat::nearest_centre_kernel<16>, grid, dim3(256), 0, st, x16, n, c16, n_centres, meta, labels); break;
        default: hipLaunchKernelGGL(flat::nearest_centre_kernel<8>, grid, dim3(256), 0, st, x16, n, c16, n_centres, meta, labels); break;
    }
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/*
 * Merge (list of L keys per query, ascending; ignored when have_list = 0) with n_sets x stride extra keys per query
 * (extra_cnt != NULL: one set, extra_cnt[q] valid entries, > stride = entries were lost -> lost[q] = 1; NULL: every entry
 * counts, sentinels allowed) into the L smallest, ascending, in place; tau[q] = min(a_(k) + 2E_q, a_(L) when the list is
 * full), +inf while the query has met fewer than k candidates.  norms_q: the queries' reference-order squared norms.
 */
int tdr_knn_flat_select_f32(uint64_t* list, int have_list, const uint64_t* extra, const int32_t* extra_cnt, int n_sets,
                            int stride, const float* norms_q, const uint32_t* meta, int64_t nq, int d, int k, int L, int terms,
                            float* tau, int32_t* lost, void* stream) {
    if (!list || !extra || !norms_q || !meta || !tau || !lost || nq <= 0 || k < 1 || L < k || n_sets < 1 || stride < 1)
        return TDR_ERR_BAD_ARG;
    if (extra_cnt && n_sets != 1) return TDR_ERR_BAD_ARG;
    if (L > 128) return TDR_ERR_UNSUPPORTED;   // the select kernel keeps two list entries per lane
    const int ks = flat::flat_scan_ks(d, terms);
    if (ks == 0) return TDR_ERR_UNSUPPORTED;
    flat::SelectParams S;
    S.list = list; S.have_list = have_list; S.extra = extra; S.extra_cnt = extra_cnt; S.n_sets = n_sets; S.stride = stride;
    S.norms_q = norms_q; S.meta = meta; S.nq = nq; S.k = k; S.L = L; S.dpad = ks * 16; S.terms = terms; S.tau = tau; S.lost = lost;
    const size_t lds = (size_t)4 * (2 * (size_t)L + (size_t)n_sets * stride) * sizeof(uint64_t);
    if (lds > 160 * 1024) return TDR_ERR_UNSUPPORTED;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flat::knn_flat_select_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(flat::knn_flat_select_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), lds, (hipStream_t)stream, S);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
