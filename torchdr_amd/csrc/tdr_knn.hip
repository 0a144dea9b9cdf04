// K1 -- exact pairwise-distance / kNN kernels for gfx950 (MI355X).
//
// Replaces the reference op sequence (all citations under /root/reference/torchdr):
//   distance/torch.py:82-91   X_norm/Y_norm + X @ Y.T + broadcast add   (ATen + MKL sgemm)
//   distance/torch.py:93-100  euclidean sqrt / angular negation
//   distance/torch.py:111-116 diagonal exclusion (+1e12)
//   utils/utils.py:215-216    kmin -> topk(k, largest=False), indices -> int32
//
// Design (CDNA4-first, not a translation):
//   * A pre-pass (pack_rows_kernel) rewrites the N x D point block into 32-row "tile images"
//     laid out exactly as the MFMA A/B fragments want them, and computes the squared norms in
//     the same summation order as ATen's CPU reduction (see oracle/knn_oracle.c).  A tile image
//     is a contiguous run of bytes, so the main kernel's HBM->LDS staging is a linear, fully
//     coalesced copy and every LDS fragment read is a conflict-free ds_read_b128.
//   * knn_scan_kernel: a workgroup of 4 wavefronts owns 128 queries; each wavefront keeps its
//     32 queries' whole feature block in VGPRs (B operand of v_mfma_f32_32x32x2_f32) for the
//     entire scan and streams database tiles (A operand) through a double-buffered LDS ring.
//     The -2 x.y contraction runs on the fp32 MFMA pipe; its result is bit-for-bit a k-ordered
//     fmaf chain, which is what MKL's sgemm produces for K <= 256 -- hence distances that are
//     bit-identical to the reference CPU backend.
//   * The operands are swapped (database rows -> MFMA rows, queries -> MFMA columns) so that a
//     lane owns ONE query: the running k-th-best threshold is a single VGPR and the hot filter
//     is one v_cmp per candidate.  The N x N tile is never written to HBM.  Survivors (rare
//     after warm-up: ~k ln(N/k) per query) are merged into a per-query k-entry list in LDS
//     ordered by the canonical (distance, index) key.
//   * Optional database split (gridDim.y) for small query counts, merged by knn_merge_kernel.
#include "tdr_common.h"
#include <stdlib.h>

namespace tdr {

constexpr int TILE_ROWS = 32;
// Empty list slot: (+inf, 0xffffffff) -- compares above every real candidate.
constexpr uint64_t KEY_SENTINEL = 0xFF800000FFFFFFFFull;

// Correctly rounded fp32 sqrt (via the fp64 root: 53 >= 2*24+2 bits makes the double rounding exact).
__device__ __forceinline__ float sqrt_rn(float x) { return (float)sqrt((double)x); }

// tile image = kq blocks of 256 floats + 32 norms + 32 floats of padding (so the norms travel as one
// 64-lane dword LDS-DMA)
__host__ __device__ __forceinline__ int64_t tile_stride_floats(int kq) { return (int64_t)kq * 256 + 64; }

// ---------------------------------------------------------------------------------------------
// ATen (AVX2 build) summation order for one "lane column": elements sq(v) = x[v*stride]^2
// for v in [0,size).  Mirrors row_sum/multi_row_sum of aten/src/ATen/native/cpu/SumKernel.cpp
// (4-way ILP rows, 4 cascade levels, level step 2^max(4, ceil_log2(size/4)/4)).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ceil_log2_i(int x) {
    if (x <= 2) return 1;
    return 32 - __clz(x - 1);
}

__device__ float aten_row_sum_sq(const float* x, int size, int stride) {
    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    const int size_ilp = size / 4;
    int lp = ceil_log2_i(size_ilp) / 4;
    if (lp < 4) lp = 4;
    const int step = 1 << lp;
    const int mask = step - 1;
    int i = 0;
    for (; i + step <= size_ilp;) {
        for (int j = 0; j < step; ++j, ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = x[(size_t)(i * 4 + r) * stride];
                acc[0][r] = __fadd_rn(acc[0][r], __fmul_rn(v, v));
            }
        }
        bool stop = false;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            if (!stop) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[j][r] = __fadd_rn(acc[j][r], acc[j - 1][r]);
                    acc[j - 1][r] = 0.f;
                }
                if ((i & (mask << (j * lp))) != 0) stop = true;
            }
        }
    }
    for (; i < size_ilp; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = x[(size_t)(i * 4 + r) * stride];
            acc[0][r] = __fadd_rn(acc[0][r], __fmul_rn(v, v));
        }
    }
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[0][r] = __fadd_rn(acc[0][r], acc[j][r]);
    for (int v = size_ilp * 4; v < size; ++v) {
        const float t = x[(size_t)v * stride];
        acc[0][0] = __fadd_rn(acc[0][0], __fmul_rn(t, t));
    }
#pragma unroll
    for (int r = 1; r < 4; ++r) acc[0][0] = __fadd_rn(acc[0][0], acc[0][r]);
    return acc[0][0];
}

// ---------------------------------------------------------------------------------------------
// pack_rows_kernel: X (n x d, row stride ldx) -> tile images + squared norms.
// Image of tile b (rows 32b .. 32b+31), kq = padded_D / 8 blocks of 256 floats:
//   img[t*256 + h*128 + i*4 + e] = X[32b + i][8t + 2e + h]      (zero outside n x d)
//   img[kq*256 + i]              = ||X[32b + i]||^2              (+inf for rows >= n)
// so that lane (h*32 + i) of a wavefront reads its four consecutive 32x32x2 operands with
// ONE 16-byte access at img + t*256 + lane*4.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ X, int64_t n, int d,
                                                        int64_t ldx, int kq, float* __restrict__ out,
                                                        float* __restrict__ norms_out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int dimg = kq * 8;
    const int ld = dimg + 8;
    const int64_t row0 = (int64_t)blockIdx.x * TILE_ROWS;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < TILE_ROWS * dimg; idx += 256) {
        const int r = idx / dimg, c = idx - r * dimg;
        float v = 0.f;
        if (row0 + r < n && c < d) v = X[(size_t)(row0 + r) * ldx + c];
        xs[r * ld + c] = v;
    }
    __syncthreads();
    float* img = out + (size_t)blockIdx.x * tile_stride_floats(kq);
    for (int idx = tid; idx < kq * 64; idx += 256) {
        const int t = idx >> 6, l = idx & 63, h = l >> 5, i = l & 31;
        const float* xr = xs + i * ld + 8 * t + h;
        f32x4 v = {xr[0], xr[2], xr[4], xr[6]};
        *reinterpret_cast<f32x4*>(img + t * 256 + l * 4) = v;
    }
    // norms: 8 lanes per row, ATen AVX2 order
    {
        const int i = tid >> 3, l = tid & 7;
        const float* xr = xs + i * ld;
        float fin;
        if (d >= 8) {
            const int vec = d / 8;
            const float p = aten_row_sum_sq(xr + l, vec, 8);
            fin = 0.f;
            if (l == 0)
                for (int k = vec * 8; k < d; ++k) fin = __fadd_rn(fin, __fmul_rn(xr[k], xr[k]));
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float ps = __shfl(p, (tid & 56) + s, 64);
                fin = __fadd_rn(fin, ps);
            }
        } else {
            fin = (l == 0) ? aten_row_sum_sq(xr, d, 1) : 0.f;
        }
        if (l == 0) {
            const bool valid = (row0 + i) < n;
            img[kq * 256 + i] = valid ? fin : __builtin_inff();
            if (valid && norms_out) norms_out[row0 + i] = fin;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Scan kernel
// ---------------------------------------------------------------------------------------------
struct KnnParams {
    const float* qp;      // packed queries
    const float* yp;      // packed database
    int64_t nq;           // number of queries
    int64_t q_offset;     // global id of query 0 (self exclusion)
    int64_t n_db;         // database rows
    int k;
    int metric;           // 0 sqeuclidean, 1 euclidean, 2 angular
    int exclude_self;
    int n_db_tiles;
    int tiles_per_split;  // database tiles per grid.y slice
    int n_splits;
    float* out_d;         // (nq, k)      when n_splits == 1
    int32_t* out_i;       // (nq, k)
    uint64_t* ws_keys;    // (n_splits, nq, k) partial keys when n_splits > 1
    int kq;               // wide kernel only: 8-dim blocks per tile image (run-time)
};

// HBM -> LDS staging of one tile image by LDS-DMA (global_load_lds: no VGPR round trip, no ds_write pass).
// The image is a linear byte run, so wave w copies the 1-KiB blocks t = w, w+4, ... (destination =
// wave-uniform base + lane*16) and wave 0 adds the 256-byte norm block with a dword-wide copy.
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int KQ, int NW>
__device__ __forceinline__ void stage_dma(const float* __restrict__ src, float* dst, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < KQ; t += NW) {
        const int blk = t + wave;
        if (blk < KQ)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + blk * 256 + lane * 4), (lptr_t)(dst + blk * 256), 16, 0, 0);
    }
    if (wave == 0)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + KQ * 256 + lane), (lptr_t)(dst + KQ * 256), 4, 0, 0);
}

// Cooperative sorted insertion: the whole wavefront inserts ONE candidate into one query's ascending
// k-entry list L (LDS, lane p owns entries p and p+64).  Every lane reads its entry and its left
// neighbour, the shifted list is written back -- no rescans, no cross-lane traffic beyond two readlanes:
// one LDS round trip.  Returns true and the new k-th key when the candidate entered the list.
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

template <int ITEMS>
__device__ __forceinline__ bool coop_insert(uint64_t* L, int k, uint64_t cand, int lane, uint64_t& new_tail) {
    uint64_t cur[ITEMS], prev[ITEMS], nv[ITEMS];
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        const int p = lane + 64 * t;
        cur[t] = (p < k) ? L[p] : KEY_SENTINEL;
        prev[t] = (p > 0 && p < k) ? L[p - 1] : 0ull;
    }
    const int tl = (k - 1) & 63, ti = (k - 1) >> 6;  // lane / item that hold the k-th entry (wave-uniform)
    uint64_t tk = 0;
#pragma unroll
    for (int t = 0; t < ITEMS; ++t)
        if (t == ti) tk = readlane_u64(cur[t], tl);
    if (cand >= tk) return false;  // wave-uniform
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        const int p = lane + 64 * t;
        nv[t] = (cur[t] < cand) ? cur[t] : ((p == 0 || prev[t] < cand) ? cand : prev[t]);
        if (p < k && nv[t] != cur[t]) L[p] = nv[t];
    }
#pragma unroll
    for (int t = 0; t < ITEMS; ++t)
        if (t == ti) new_tail = readlane_u64(nv[t], tl);
    return true;
}

// ---- one tile step of the scan (software pipelined) -----------------------------------------------------
// A wavefront owns QB blocks of 32 queries (QB = 1: 2 wavefronts per SIMD; QB = 2: one wavefront per SIMD that
// drives two independent MFMA chains off the same A fragments -- half the LDS reads, barriers and staging
// per matrix instruction, and 256 queries per workgroup halve the L2 traffic).
// tile_step multiplies tile T into `acc` (one k-ordered MFMA chain per query block) and, IN THE SAME
// INSTRUCTION STREAM, finishes tile T-1 out of `accp`: its candidate distances are formed between the MFMAs
// of tile T (an in-order wavefront overlaps VALU with a 64-cycle MFMA only if the VALU sits between MFMAs in
// program order), folded into a running minimum and tested against the lane's k-th-best threshold with ONE
// compare.  Only when some lane has a survivor is the scalar-driven insertion loop entered.
template <int QB>
struct ScanCtx {
    const KnnParams* P;
    uint64_t* keys;   // this wave's lists [QB][32][k]
    int k, lane, q, h;
    int64_t qt0;      // first query tile of this wave (tiles qt0 .. qt0 + QB - 1)
    float xn[QB];
    bool angular;
};

// Rare path.  For each candidate slot the compare against the threshold IS the ballot (v_cmp writes the
// 64-lane mask to SGPRs); slots without a survivor cost one scalar test.  Survivors are walked from the
// scalar unit: value and row index are wave-uniform (readlane), only the list update itself is vector work.
// tau_d is refreshed after every insertion, so later slots filter with the new threshold.
template <int ITEMS, int QB>
__device__ __forceinline__ void scan_insert(const ScanCtx<QB>& C, const float (&dv)[QB][16], const float (&pmin)[QB][4],
                                            int Tprev, float (&tau_d)[QB]) {
    const KnnParams& P = *C.P;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (!__any(pmin[qb][g] <= tau_d[qb])) continue;  // quarter without survivors: one scalar test
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                unsigned long long m = __ballot(dv[qb][r] <= tau_d[qb]);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const int sq = src & 31;
                    const int64_t j = (int64_t)Tprev * 32 + 4 * (src >> 5) + e + 8 * g;
                    if (j >= P.n_db || (P.exclude_self && j == (C.qt0 + qb) * 32 + sq + P.q_offset)) continue;
                    const float dval =
                        __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dv[qb][r]), src));
                    uint64_t new_tail;
                    if (coop_insert<ITEMS>(C.keys + ((size_t)qb * 32 + sq) * C.k, C.k, mkkey(dval, (uint32_t)j), C.lane,
                                           new_tail)) {
                        if (C.q == sq) tau_d[qb] = u2f((uint32_t)(new_tail >> 32));
                    }
                }
            }
        }
    }
}

// candidate values of one finished tile: c = (||x||^2 + ||y||^2) - 2 x.y  (distance/torch.py:91); rows
// (r&3) + 8*(r>>2) + 4*h of the tile.  `g` selects which quarter (4 values) to form -- the caller spreads
// the four quarters over the MFMA groups of the next tile.
template <int QB>
__device__ __forceinline__ void form_part(const ScanCtx<QB>& C, const f32x16 (&accp)[QB], const float* ynp, int g,
                                          float (&dv)[QB][16], float (&pmin)[QB][4]) {
    const f32x4 y4 = *reinterpret_cast<const f32x4*>(ynp + 8 * g);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float m = __builtin_inff();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            float c;
            if (C.angular) c = -accp[qb][r];
            else c = __builtin_fmaf(-2.0f, accp[qb][r], __fadd_rn(C.xn[qb], y4[e]));  // 2*acc exact: same rounding as s - 2*acc
            dv[qb][r] = c;
            m = fminf(m, c);
        }
        pmin[qb][g] = m;  // minimum of this quarter: the rare path only scans quarters that hold a survivor
    }
}

template <int QB>
__device__ __forceinline__ bool any_survivor(const float (&pmin)[QB][4], const float (&tau_d)[QB]) {
    bool hit = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
        hit |= (fminf(fminf(pmin[qb][0], pmin[qb][1]), fminf(pmin[qb][2], pmin[qb][3])) <= tau_d[qb]);
    return __any(hit);
}

template <int KQ, int ITEMS, int QB, bool HAVE_PREV>
__device__ __forceinline__ void tile_step(const ScanCtx<QB>& C, const float* __restrict__ img, const float (&b)[QB][4 * KQ],
                                          f32x16 (&acc)[QB], const f32x16 (&accp)[QB], const float* ynp_prev, int Tprev,
                                          float (&tau_d)[QB]) {
    constexpr int GQ = (KQ >= 4) ? 4 : KQ, NG = KQ / GQ;
    constexpr int PARTS_PER_GROUP = (4 + NG - 1) / NG;  // spread the 4 quarters of the previous tile over the groups
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[qb][r] = 0.f;
    float dv[QB][16];
    float pmin[QB][4];
    const float* ap = img + C.lane * 4;
    f32x4 a0[GQ], a1[GQ];
#pragma unroll
    for (int u = 0; u < GQ; ++u) a0[u] = *reinterpret_cast<const f32x4*>(ap + u * 256);
    int part = 0;
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
        if (g + 1 < NG) {
#pragma unroll
            for (int u = 0; u < GQ; ++u) a1[u] = *reinterpret_cast<const f32x4*>(ap + ((g + 1) * GQ + u) * 256);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMA group
        if (HAVE_PREV) {
#pragma unroll
            for (int pp = 0; pp < PARTS_PER_GROUP; ++pp)
                if (part + pp < 4) form_part<QB>(C, accp, ynp_prev, part + pp, dv, pmin);
        }
        part += PARTS_PER_GROUP;
#pragma unroll
        for (int u = 0; u < GQ; ++u) {
            const int t = g * GQ + u;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][e], b[qb][4 * t + e], acc[qb], 0, 0, 0);
        }
        if (g + 1 < NG) {
            if (g + 2 < NG) {
#pragma unroll
                for (int u = 0; u < GQ; ++u) a0[u] = *reinterpret_cast<const f32x4*>(ap + ((g + 2) * GQ + u) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (HAVE_PREV) {
#pragma unroll
                for (int pp = 0; pp < PARTS_PER_GROUP; ++pp)
                    if (part + pp < 4) form_part<QB>(C, accp, ynp_prev, part + pp, dv, pmin);
            }
            part += PARTS_PER_GROUP;
#pragma unroll
            for (int u = 0; u < GQ; ++u) {
                const int t = (g + 1) * GQ + u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u][e], b[qb][4 * t + e], acc[qb], 0, 0, 0);
            }
        }
    }
    if (HAVE_PREV) {
#ifdef TDR_ABLATE_NOINSERT
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int g = 0; g < 4; ++g) { asm volatile("" ::"v"(pmin[qb][g])); pmin[qb][g] = __builtin_inff(); }
#endif
        if (any_survivor<QB>(pmin, tau_d)) scan_insert<ITEMS, QB>(C, dv, pmin, Tprev, tau_d);
    }
}

// finish the last tile (no next tile to hide it behind)
template <int ITEMS, int QB>
__device__ __forceinline__ void tile_drain(const ScanCtx<QB>& C, const f32x16 (&accp)[QB], const float* ynp_prev, int Tprev,
                                           float (&tau_d)[QB]) {
    float dv[QB][16];
    float pmin[QB][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) form_part<QB>(C, accp, ynp_prev, g, dv, pmin);
#ifdef TDR_ABLATE_NOINSERT
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int g = 0; g < 4; ++g) pmin[qb][g] = __builtin_inff();
#endif
    if (any_survivor<QB>(pmin, tau_d)) scan_insert<ITEMS, QB>(C, dv, pmin, Tprev, tau_d);
}

template <int KQ, int ITEMS, int QB>
__global__ __launch_bounds__(256, (QB == 2) ? 1 : 2) void knn_scan_kernel(const KnnParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NW = 4;
    constexpr int TILE_F = KQ * 256 + 64;
    constexpr int IMG_F = KQ * 256;  // the LDS copy holds the blocks only; norms go to the ring below
    float* tile0 = reinterpret_cast<float*>(smem_raw);
    float* tile1 = tile0 + IMG_F;
    float* nring = tile1 + IMG_F;                                          // [4 slots][64 floats]
    uint64_t* keys_all = reinterpret_cast<uint64_t*>(nring + 4 * 64);      // [NW waves][QB][32 queries][k] ascending
    const int k = P.k;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane & 31;
    const int h = lane >> 5;
    uint64_t* keys = keys_all + (size_t)wave * QB * k * 32;

    const int64_t n_qtiles = (P.nq + 31) / 32;
    const int64_t qt0 = ((int64_t)blockIdx.x * NW + wave) * QB;
    const bool wave_active = qt0 < n_qtiles;

    // --- query blocks -> registers (B operands), once per workgroup lifetime
    float b[QB][4 * KQ];
    ScanCtx<QB> C;
    float tau_d[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int64_t qt = qt0 + qb;
        const bool blk_active = qt < n_qtiles;
        if (blk_active) {
            const float* qimg = P.qp + (size_t)qt * TILE_F;
#pragma unroll
            for (int t = 0; t < KQ; ++t) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(qimg + t * 256 + lane * 4);
                b[qb][4 * t + 0] = v[0]; b[qb][4 * t + 1] = v[1]; b[qb][4 * t + 2] = v[2]; b[qb][4 * t + 3] = v[3];
            }
            C.xn[qb] = qimg[KQ * 256 + q];
        } else {
#pragma unroll
            for (int t = 0; t < 4 * KQ; ++t) b[qb][t] = 0.f;
            C.xn[qb] = 0.f;
        }
        const bool lane_valid = blk_active && (qt * 32 + q < P.nq);
        tau_d[qb] = lane_valid ? __builtin_inff() : -__builtin_inff();
    }
    for (int p = lane; p < QB * k * 32; p += 64) keys[p] = KEY_SENTINEL;

    const int split = blockIdx.y;
    const int t_begin = split * P.tiles_per_split;
    int t_end = t_begin + P.tiles_per_split;
    if (t_end > P.n_db_tiles) t_end = P.n_db_tiles;

    C.P = &P; C.keys = keys; C.k = k; C.lane = lane; C.q = q; C.h = h; C.qt0 = qt0;
    C.angular = (P.metric == 2);

    // stage(T): tile image -> tile[(T - t_begin) & 1], norms -> nring[(T - t_begin) & 3]
    auto stage = [&](int T) {
        const int rel = T - t_begin;
        const float* src = P.yp + (size_t)T * TILE_F;
        float* dst = (rel & 1) ? tile1 : tile0;
#pragma unroll
        for (int t = 0; t < KQ; t += NW) {
            const int blk = t + wave;
            if (blk < KQ)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + blk * 256 + lane * 4), (lptr_t)(dst + blk * 256), 16, 0, 0);
        }
        if (wave == NW - 1)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + KQ * 256 + lane), (lptr_t)(nring + (rel & 3) * 64), 4, 0, 0);
    };
    if (t_begin < t_end) stage(t_begin);
    __syncthreads();

    f32x16 accA[QB], accB[QB];
    int T = t_begin;
#define TDR_YN(Tx) (nring + (((Tx) - t_begin) & 3) * 64 + 4 * h)
    // first tile: nothing to finish yet
    if (T < t_end) {
#ifndef TDR_ABLATE_NOSTAGE
        if (T + 1 < t_end) stage(T + 1);
#endif
        if (wave_active) tile_step<KQ, ITEMS, QB, false>(C, tile0, b, accA, accA, nring, T, tau_d);
#ifndef TDR_ABLATE_NOBARRIER
        __syncthreads();
#endif
        ++T;
    }
    // steady state, unrolled by two so that the accumulators / buffers are static
    while (T < t_end) {
        {   // odd relative tile: image in tile1, finishes the tile held in accA
#ifndef TDR_ABLATE_NOSTAGE
            if (T + 1 < t_end) stage(T + 1);
#endif
            if (wave_active) tile_step<KQ, ITEMS, QB, true>(C, tile1, b, accB, accA, TDR_YN(T - 1), T - 1, tau_d);
#ifndef TDR_ABLATE_NOBARRIER
            __syncthreads();
#endif
            ++T;
        }
        if (T < t_end) {  // even relative tile: image in tile0, finishes accB
#ifndef TDR_ABLATE_NOSTAGE
            if (T + 1 < t_end) stage(T + 1);
#endif
            if (wave_active) tile_step<KQ, ITEMS, QB, true>(C, tile0, b, accA, accB, TDR_YN(T - 1), T - 1, tau_d);
#ifndef TDR_ABLATE_NOBARRIER
            __syncthreads();
#endif
            ++T;
        } else {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) accA[qb] = accB[qb];  // the last computed tile is always drained from accA
        }
    }
    if (wave_active && t_begin < t_end) tile_drain<ITEMS, QB>(C, accA, TDR_YN(t_end - 1), t_end - 1, tau_d);
#undef TDR_YN

    // --- emit: every list is already ascending by (distance, index)
    if (wave_active) {
        for (int jq = 0; jq < 32 * QB; ++jq) {
            const int64_t qi = qt0 * 32 + jq;
            if (qi >= P.nq) break;
            for (int p = lane; p < k; p += 64) {
                const uint64_t mine = keys[(size_t)jq * k + p];
                if (P.n_splits > 1) {
                    P.ws_keys[((size_t)split * P.nq + qi) * k + p] = mine;
                } else {
                    float c = u2f((uint32_t)(mine >> 32));
                    if (P.metric == 1) c = sqrt_rn(fmaxf(c, 0.f));
                    P.out_d[(size_t)qi * k + p] = c;
                    P.out_i[(size_t)qi * k + p] = (int32_t)(uint32_t)(mine & 0xffffffffu);
                }
            }
        }
    }
}

// Merge the per-split sorted partial lists: one wavefront per query, rank by counting.
__global__ __launch_bounds__(256) void knn_merge_kernel(const uint64_t* __restrict__ ws, int64_t nq, int k,
                                                        int n_splits, int metric, float* __restrict__ out_d,
                                                        int32_t* __restrict__ out_i) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
    const int total = n_splits * k;
    uint64_t* mybuf = buf + (size_t)wave * total;
    if (qi < nq)
        for (int p = lane; p < total; p += 64) {
            const int s = p / k, r = p - s * k;
            mybuf[p] = ws[((size_t)s * nq + qi) * k + r];
        }
    __syncthreads();
    if (qi >= nq) return;
    for (int p0 = 0; p0 < total; p0 += 64) {
        const int p = p0 + lane;
        const uint64_t mine = (p < total) ? mybuf[p] : KEY_SENTINEL;
        int rank = 0;
        for (int pp = 0; pp < total; ++pp) rank += (mybuf[pp] < mine) ? 1 : 0;
        if (p < total && rank < k && mine != KEY_SENTINEL) {
            float c = u2f((uint32_t)(mine >> 32));
            if (metric == 1) c = sqrt_rn(fmaxf(c, 0.f));
            out_d[(size_t)qi * k + rank] = c;
            out_i[(size_t)qi * k + rank] = (int32_t)(uint32_t)(mine & 0xffffffffu);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dense distances (k = None path, distance/torch.py:91-116 without kmin): same MFMA core, the
// tile is written out instead of filtered.  diag_add is added to C[i][i] when exclude_self.
// ---------------------------------------------------------------------------------------------
template <int KQ>
__global__ __launch_bounds__(256, 2) void dense_dist_kernel(const float* __restrict__ qp,
                                                            const float* __restrict__ yp, int64_t nq,
                                                            int64_t q_offset, int64_t n_db, int metric,
                                                            int exclude_self, float diag_add,
                                                            float* __restrict__ out, int64_t ldo) {
    constexpr int TILE_F = KQ * 256 + 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 31, h = lane >> 5;
    const int64_t qt = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n_qtiles = (nq + 31) / 32;
    if (qt >= n_qtiles) return;
    const int64_t T = blockIdx.y;  // database tile
    const float* qimg = qp + (size_t)qt * TILE_F;
    const float* img = yp + (size_t)T * TILE_F;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < KQ; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(img + t * 256 + lane * 4);
        const f32x4 bq = *reinterpret_cast<const f32x4*>(qimg + t * 256 + lane * 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bq[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], bq[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], bq[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], bq[3], acc, 0, 0, 0);
    }
    const float xn = qimg[KQ * 256 + q];
    const int64_t gq = qt * 32 + q;
    if (gq >= nq) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int64_t j = T * 32 + i;
        if (j >= n_db) continue;
        float c;
        if (metric == 2) c = -acc[r];
        else {
            c = __fsub_rn(__fadd_rn(xn, img[KQ * 256 + i]), __fmul_rn(2.0f, acc[r]));
            if (metric == 1) c = sqrt_rn(fmaxf(c, 0.f));
        }
        if (exclude_self && j == gq + q_offset) c = __fadd_rn(c, diag_add);
        out[(size_t)gq * ldo + j] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// kNN-set overlap (eval/neighborhood_preservation.py:175-181): out[i] = |{p : a[i][p] in b[i][:]}| / K.
// One wavefront per row; the b row sits in LDS and is read as a broadcast.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_overlap_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                          int64_t n, int K, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t* bs = reinterpret_cast<int32_t*>(smem_raw) + (size_t)wave * K;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= n) return;
    for (int p = lane; p < K; p += 64) bs[p] = b[(size_t)row * K + p];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int cnt = 0;
    for (int p = lane; p < K; p += 64) {
        const int32_t ai = a[(size_t)row * K + p];
        bool hit = false;
        for (int qq = 0; qq < K; ++qq) hit |= (bs[qq] == ai);
        cnt += hit ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) out[row] = (float)cnt / (float)K;
}

// ---------------------------------------------------------------------------------------------
// General feature dimension (D > 256): the contraction is a plain library GEMM on a (queries x database chunk)
// block (the host calls rocBLAS through torch.mm); this kernel is the rest of distance/torch.py:91-120 -- forms
// c = (||x||^2 + ||y||^2) - 2 G (or -G), excludes the query itself and folds the block into each query's running
// ascending k-list (same 64-bit (distance, index) keys and cooperative insertion as the scan kernel).  One wavefront
// per query row; the list lives in LDS during the launch and in `run_keys` between database chunks.
// ---------------------------------------------------------------------------------------------
// distance/torch.py:101-107: arccosh(1 + 2 relu(C) / ((1 - |x|^2)(1 - |y|^2)) + 1e-8)^2 in the reference's fp32 op order
__device__ __forceinline__ float sqhyperbolic_from(float c_sq, float xn, float yn) {
    const float C = fmaxf(c_sq, 0.f);
    const float den = __fmul_rn(__fsub_rn(1.0f, xn), __fsub_rn(1.0f, yn));
    const float w = __fadd_rn(__fadd_rn(1.0f, __fmul_rn(2.0f, __fdiv_rn(C, den))), 1e-8f);
    const float u = acoshf(w);
    return __fmul_rn(u, u);
}

struct TopkMergeParams {
    const float* G;        // (nq, nd) block of X Y^T, row stride ldg
    int64_t ldg;
    int64_t nq, nd;
    const float* xn;       // (nq) squared norms of the queries (unused for angular)
    const float* yn;       // (nd) squared norms of this database chunk
    int64_t q_global0;     // global index of query 0 (self exclusion)
    int64_t d_global0;     // global index of database row 0 of this chunk
    int k, metric, exclude_self;
    uint64_t* run_keys;    // (nq, k) ascending, KEY_SENTINEL padded
    const int32_t* cand;   // optional (nq, nd) database index of every column (row stride ldg); entries < 0 are skipped
};

template <int ITEMS>
__global__ __launch_bounds__(256) void topk_merge_kernel(const TopkMergeParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
    if (qi >= P.nq) return;
    uint64_t* Lst = reinterpret_cast<uint64_t*>(smem_raw) + (size_t)wave * P.k;
    for (int p = lane; p < P.k; p += 64) Lst[p] = P.run_keys[(size_t)qi * P.k + p];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float tau = u2f((uint32_t)(Lst[P.k - 1] >> 32));  // k-th best so far (+inf while the list is not full)
    const float xq = (P.metric == 2 || P.metric == 3) ? 0.f : P.xn[qi];
    const float* g = P.G + (size_t)qi * P.ldg;
    const int64_t self_j = P.exclude_self ? (P.q_global0 + qi - P.d_global0) : -1;
    for (int64_t j0 = 0; j0 < P.nd; j0 += 64) {
        const int64_t j = j0 + lane;
        float c = __builtin_inff();
        uint32_t id = (uint32_t)(P.d_global0 + j);
        bool live = j < P.nd && j != self_j;
        if (live && P.cand) {
            const int32_t ci = P.cand[(size_t)qi * P.ldg + j];
            live = ci >= 0;
            id = (uint32_t)ci;
        }
        if (live) {
            const float gv = g[j];
            if (P.metric == 3) c = gv;
            else if (P.metric == 2) c = -gv;
            else {
                const float yj = P.yn[j];
                c = __builtin_fmaf(-2.0f, gv, __fadd_rn(xq, yj));
                if (P.metric == 4) c = sqhyperbolic_from(c, xq, yj);
            }
        }
        unsigned long long m = __ballot(c <= tau);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const float cv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), src));
            const uint32_t iv = (uint32_t)__builtin_amdgcn_readlane((int)id, src);
            uint64_t new_tail;
            if (coop_insert<ITEMS>(Lst, P.k, mkkey(cv, iv), lane, new_tail))
                tau = u2f((uint32_t)(new_tail >> 32));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int p = lane; p < P.k; p += 64) P.run_keys[(size_t)qi * P.k + p] = Lst[p];
}

// run_keys -> (distance, index) outputs (sqrt for the euclidean metric)
__global__ __launch_bounds__(256) void topk_emit_kernel(const uint64_t* __restrict__ keys, int64_t total, int metric,
                                                        float* __restrict__ out_d, int32_t* __restrict__ out_i) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint64_t key = keys[i];
    float c = u2f((uint32_t)(key >> 32));
    if (metric == 1) c = sqrt_rn(fmaxf(c, 0.f));
    out_d[i] = c;
    out_i[i] = (int32_t)(uint32_t)(key & 0xffffffffu);
}

// Gram block -> sqhyperbolic distances, in place (dense form of distance/torch.py:101-107)
__global__ __launch_bounds__(256) void hyperbolic_epilogue_kernel(float* __restrict__ G, int64_t ld, int64_t nq, int64_t nd,
                                                                 const float* __restrict__ xn, const float* __restrict__ yn) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= nd || i >= nq) return;
    const float xq = xn[i], yj = yn[j];
    float* g = G + (size_t)i * ld + j;
    *g = sqhyperbolic_from(__builtin_fmaf(-2.0f, *g, __fadd_rn(xq, yj)), xq, yj);
}

__global__ __launch_bounds__(256) void fill_keys_kernel(uint64_t* __restrict__ keys, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) keys[i] = KEY_SENTINEL;
}


// ---------------------------------------------------------------------------------------------
// Wide rows (D > 256): the query block no longer fits the register file, so the contraction is
// K-chunked.  A wavefront still owns 32 queries and a lane ONE query (same filter / k-list
// machinery as knn_scan_kernel), but it accumulates a GROUP of WIDE_TG database tiles at once:
// per K step of WIDE_KC blocks (32 dims) it reads its query fragment once (4 x 16 B per lane,
// from L2 / the infinity cache) and multiplies it into the WIDE_TG accumulators with A fragments
// taken from a double-buffered LDS ring the four wavefronts stage together by LDS-DMA.  After the
// last K step the group's 4 x 32 x 32 candidate values are formed, filtered against the lane's
// k-th best and merged -- the N x N block never exists in memory (the library-GEMM form of
// distance/torch.py:91 wrote 1 GiB blocks of X Y^T that a second kernel re-read).
// Each output element is one k-ordered fma chain over the whole row, as in the D <= 256 kernel.
// ---------------------------------------------------------------------------------------------
constexpr int WIDE_TG = 4;  // database tiles accumulated together (= wavefronts: wave w stages tile w of the group)
constexpr int WIDE_KC = 4;  // 8-dim blocks per K step

__global__ __launch_bounds__(256) void pack_wide_kernel(const float* __restrict__ X, int64_t n, int d, int64_t ldx, int kq,
                                                        float* __restrict__ out, float* __restrict__ norms_out) {
    __shared__ float xs[TILE_ROWS][264];
    const int64_t row0 = (int64_t)blockIdx.x * TILE_ROWS;
    const int tid = threadIdx.x;
    float* img = out + (size_t)blockIdx.x * tile_stride_floats(kq);
    for (int c0 = 0; c0 < kq * 8; c0 += 256) {
        for (int idx = tid; idx < TILE_ROWS * 256; idx += 256) {
            const int r = idx >> 8, c = idx & 255;
            float v = 0.f;
            if (row0 + r < n && c0 + c < d) v = X[(size_t)(row0 + r) * ldx + c0 + c];
            xs[r][c] = v;
        }
        __syncthreads();
        const int t0 = c0 / 8;
        const int nblk = (kq - t0 < 32) ? kq - t0 : 32;
        for (int idx = tid; idx < nblk * 64; idx += 256) {
            const int t = idx >> 6, l = idx & 63, h = l >> 5, i = l & 31;
            const float* xr = &xs[i][8 * t + h];
            f32x4 v = {xr[0], xr[2], xr[4], xr[6]};
            *reinterpret_cast<f32x4*>(img + (size_t)(t0 + t) * 256 + l * 4) = v;
        }
        __syncthreads();
    }
    // norms: 8 lanes per row in ATen's AVX2 order (as pack_rows_kernel), read from the source rows
    {
        const int i = tid >> 3, l = tid & 7;
        const bool valid = (row0 + i) < n;
        const float* xr = X + (size_t)(valid ? row0 + i : 0) * ldx;
        const int vec = d / 8;
        const float p = aten_row_sum_sq(xr + l, vec, 8);
        float fin = 0.f;
        if (l == 0)
            for (int k = vec * 8; k < d; ++k) fin = __fadd_rn(fin, __fmul_rn(xr[k], xr[k]));
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float ps = __shfl(p, (tid & 56) + s, 64);
            fin = __fadd_rn(fin, ps);
        }
        if (l == 0) {
            img[(size_t)kq * 256 + i] = valid ? fin : __builtin_inff();
            img[(size_t)kq * 256 + 32 + i] = 0.f;
            if (valid && norms_out) norms_out[row0 + i] = fin;
        }
    }
}

// dense distance tiles for wide rows: dense_dist_kernel with a run-time number of 8-feature blocks
__global__ __launch_bounds__(256, 2) void dense_wide_kernel(const float* __restrict__ qp, const float* __restrict__ yp, int64_t nq,
                                                            int64_t q_offset, int64_t n_db, int kq, int metric, int exclude_self,
                                                            float diag_add, float* __restrict__ out, int64_t ldo) {
    const int64_t TILE_F = tile_stride_floats(kq);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 31, h = lane >> 5;
    const int64_t qt = (int64_t)blockIdx.x * 4 + wave;
    if (qt >= (nq + 31) / 32) return;
    const int64_t T = blockIdx.y;
    const float* qimg = qp + (size_t)qt * TILE_F;
    const float* img = yp + (size_t)T * TILE_F;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int t0 = 0; t0 < kq; t0 += 4) {  // kq is a multiple of 4
        f32x4 a[4], bq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(img + (size_t)(t0 + u) * 256 + lane * 4);
            bq[u] = *reinterpret_cast<const f32x4*>(qimg + (size_t)(t0 + u) * 256 + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], bq[u][e], acc, 0, 0, 0);
    }
    const float xn = qimg[(size_t)kq * 256 + q];
    const int64_t gq = qt * 32 + q;
    if (gq >= nq) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int64_t j = T * 32 + i;
        if (j >= n_db) continue;
        float c;
        if (metric == 2) c = -acc[r];
        else {
            c = __fsub_rn(__fadd_rn(xn, img[(size_t)kq * 256 + i]), __fmul_rn(2.0f, acc[r]));
            if (metric == 1) c = sqrt_rn(fmaxf(c, 0.f));
        }
        if (exclude_self && j == gq + q_offset) c = __fadd_rn(c, diag_add);
        out[(size_t)gq * ldo + j] = c;
    }
}

template <int ITEMS>
__global__ __launch_bounds__(256, 2) void knn_wide_kernel(const KnnParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NW = 4, TG = WIDE_TG, KC = WIDE_KC;
    constexpr int BUF_F = TG * KC * 256;
    float* buf0 = reinterpret_cast<float*>(smem_raw);
    float* buf1 = buf0 + BUF_F;
    float* nring = buf1 + BUF_F;                                          // [2 groups][TG tiles][64 floats]
    uint64_t* keys_all = reinterpret_cast<uint64_t*>(nring + 2 * TG * 64);  // [NW waves][32 queries][k] ascending
    const int k = P.k, kq = P.kq;
    const int64_t TILE_F = tile_stride_floats(kq);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane & 31, h = lane >> 5;
    uint64_t* keys = keys_all + (size_t)wave * k * 32;
    const int64_t n_qtiles = (P.nq + 31) / 32;
    const int64_t qt0 = (int64_t)blockIdx.x * NW + wave;
    const bool wave_active = qt0 < n_qtiles;
    const float* qimg = P.qp + (size_t)(wave_active ? qt0 : 0) * TILE_F;

    ScanCtx<1> C;
    float tau_d[1];
    C.xn[0] = wave_active ? qimg[(size_t)kq * 256 + q] : 0.f;
    tau_d[0] = (wave_active && qt0 * 32 + q < P.nq) ? __builtin_inff() : -__builtin_inff();
    for (int p = lane; p < k * 32; p += 64) keys[p] = KEY_SENTINEL;
    C.P = &P; C.keys = keys; C.k = k; C.lane = lane; C.q = q; C.h = h; C.qt0 = qt0;
    C.angular = (P.metric == 2);

    const int split = blockIdx.y;
    const int t_begin = split * P.tiles_per_split;
    int t_end = t_begin + P.tiles_per_split;
    if (t_end > P.n_db_tiles) t_end = P.n_db_tiles;
    const int spg = kq / KC;  // K steps per group
    const int n_groups = (t_end > t_begin) ? (t_end - t_begin + TG - 1) / TG : 0;
    const int total = n_groups * spg;

    auto stage = [&](int step) {
        const int g = step / spg, c = step - g * spg;
        int T = t_begin + g * TG + wave;
        if (T >= t_end) T = t_end - 1;  // short last group: a copy of the last tile, never merged
        const float* src = P.yp + (size_t)T * TILE_F;
        float* dst = ((step & 1) ? buf1 : buf0) + wave * KC * 256;
#pragma unroll
        for (int u = 0; u < KC; ++u)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)(c * KC + u) * 256 + lane * 4), (lptr_t)(dst + u * 256), 16, 0, 0);
        if (c == 0)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kq * 256 + lane), (lptr_t)(nring + ((g & 1) * TG + wave) * 64), 4, 0, 0);
    };
    f32x4 bcur[KC], bnext[KC];
    auto load_b = [&](int c, f32x4 (&b)[KC]) {
#pragma unroll
        for (int u = 0; u < KC; ++u) b[u] = *reinterpret_cast<const f32x4*>(qimg + (size_t)(c * KC + u) * 256 + lane * 4);
    };
    if (total > 0) { stage(0); load_b(0, bcur); }
    __syncthreads();

    f32x16 acc[TG];
    int g = 0, c = 0;
    for (int step = 0; step < total; ++step) {
        if (step + 1 < total) {
            stage(step + 1);
            load_b((c + 1 == spg) ? 0 : c + 1, bnext);
        }
        if (c == 0) {
#pragma unroll
            for (int j = 0; j < TG; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
        const float* ap = ((step & 1) ? buf1 : buf0) + lane * 4;
#pragma unroll
        for (int j = 0; j < TG; ++j) {
            f32x4 a[KC];
#pragma unroll
            for (int u = 0; u < KC; ++u) a[u] = *reinterpret_cast<const f32x4*>(ap + (j * KC + u) * 256);
#pragma unroll
            for (int u = 0; u < KC; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], bcur[u][e], acc[j], 0, 0, 0);
        }
        if (c == spg - 1) {
            if (wave_active) {
#pragma unroll
                for (int j = 0; j < TG; ++j) {
                    const int T = t_begin + g * TG + j;
                    if (T < t_end) {
                        f32x16 one[1];
                        one[0] = acc[j];
                        float dv[1][16], pmin[1][4];
                        const float* ynp = nring + ((g & 1) * TG + j) * 64 + 4 * h;
#pragma unroll
                        for (int part = 0; part < 4; ++part) form_part<1>(C, one, ynp, part, dv, pmin);
                        if (any_survivor<1>(pmin, tau_d)) scan_insert<ITEMS, 1>(C, dv, pmin, T, tau_d);
                    }
                }
            }
            c = 0; ++g;
        } else {
            ++c;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < KC; ++u) bcur[u] = bnext[u];
    }

    if (wave_active) {
        for (int jq = 0; jq < 32; ++jq) {
            const int64_t qi = qt0 * 32 + jq;
            if (qi >= P.nq) break;
            for (int p = lane; p < k; p += 64) {
                const uint64_t mine = keys[(size_t)jq * k + p];
                if (P.n_splits > 1) {
                    P.ws_keys[((size_t)split * P.nq + qi) * k + p] = mine;
                } else {
                    float cc = u2f((uint32_t)(mine >> 32));
                    if (P.metric == 1) cc = sqrt_rn(fmaxf(cc, 0.f));
                    P.out_d[(size_t)qi * k + p] = cc;
                    P.out_i[(size_t)qi * k + p] = (int32_t)(uint32_t)(mine & 0xffffffffu);
                }
            }
        }
    }
}

static inline int pick_kq(int d) {
    if (d <= 32) return 4;
    if (d <= 64) return 8;
    if (d <= 128) return 16;
    if (d <= 256) return 32;
    return 0;
}

}  // namespace tdr

using namespace tdr;

// Workgroup shape: 4 wavefronts x QB query blocks of 32 (QB = 1: 128 queries, 2 workgroups per CU;
// QB = 2: 256 queries, 1 workgroup per CU, two MFMA chains per wavefront).
static size_t knn_lds_bytes(int kq, int k, int qb) {
    return (size_t)2 * (kq * 256) * sizeof(float) + (size_t)4 * 64 * sizeof(float) +
           (size_t)4 * qb * k * 32 * sizeof(uint64_t);
}

static int knn_qb_pref() { return 1; }  // QB = 2 measured slower (115 vs 126 TFLOP/s); kept instantiated for ablations

static int knn_qb(int kq, int k) {
    int qb = knn_qb_pref();
    if (qb == 2 && (kq > 16 || knn_lds_bytes(kq, k, 2) > 160 * 1024)) qb = 1;
    return qb;
}

template <int KQ, int ITEMS, int QB>
static int launch_scan_w(const KnnParams& P, int n_wgs, size_t lds, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_scan_kernel<KQ, ITEMS, QB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((knn_scan_kernel<KQ, ITEMS, QB>), dim3((unsigned)n_wgs, (unsigned)P.n_splits), dim3(256), lds, st, P);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}
template <int KQ, int ITEMS>
static int launch_scan_i(const KnnParams& P, int n_wgs, size_t lds, int qb, hipStream_t st) {
    if constexpr (KQ <= 16) {
        if (qb == 2) return launch_scan_w<KQ, ITEMS, 2>(P, n_wgs, lds, st);
    }
    return launch_scan_w<KQ, ITEMS, 1>(P, n_wgs, lds, st);
}
template <int KQ>
static int launch_scan(const KnnParams& P, int n_wgs, size_t lds, int qb, hipStream_t st) {
    if (P.k <= 64) return launch_scan_i<KQ, 1>(P, n_wgs, lds, qb, st);
    return launch_scan_i<KQ, 2>(P, n_wgs, lds, qb, st);
}

constexpr int TOPK_MERGE_MAX_K = 1024;   // 16 list entries per lane, 4 x 8 KiB of LDS per workgroup

static int launch_topk_merge(const TopkMergeParams& P, void* stream) {
    const int k = P.k;
    const int64_t nq = P.nq;
    const size_t lds = (size_t)4 * k * sizeof(uint64_t);
    const dim3 grid((unsigned)((nq + 3) / 4));
    if (k <= 64) hipLaunchKernelGGL(topk_merge_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, P);
    else if (k <= 128) hipLaunchKernelGGL(topk_merge_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, P);
    else if (k <= 256) hipLaunchKernelGGL(topk_merge_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, P);
    else if (k <= 512) hipLaunchKernelGGL(topk_merge_kernel<8>, grid, dim3(256), lds, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(topk_merge_kernel<16>, grid, dim3(256), lds, (hipStream_t)stream, P);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}


extern "C" {

// Number of floats needed for the packed image of n rows of dimension d (0 if d unsupported).
int64_t tdr_packed_floats(int64_t n, int d) {
    const int kq = pick_kq(d);
    if (kq == 0 || n < 0) return 0;
    const int64_t tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
    return tiles * tile_stride_floats(kq);
}

int tdr_pack_rows_f32(const float* X, int64_t n, int d, int64_t ldx, float* packed, float* norms_out,
                      void* stream) {
    if (!X || !packed || n <= 0 || d <= 0 || ldx < d) return TDR_ERR_BAD_ARG;
    const int kq = pick_kq(d);
    if (kq == 0) return TDR_ERR_UNSUPPORTED;
    const int64_t tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
    const size_t shmem = (size_t)TILE_ROWS * (kq * 8 + 8) * sizeof(float);
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)tiles), dim3(256), shmem, (hipStream_t)stream, X, n, d,
                       ldx, kq, packed, norms_out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

// Launch plan.  A workgroup owns 128 queries and two workgroups fit a CU, so the chip runs `slots` of them
// at a time and a launch of W workgroups takes ceil(W / slots) rounds.  (1) Few queries: slice the
// database over gridDim.y so that >= ~2 rounds of workgroups exist.  (2) Many queries: the main launch
// covers an exact multiple of `slots` workgroups and the leftover query tiles (the partial last round)
// go to a second launch with the database sliced, so that round is spread over every CU too.
static int device_slots() {
    static int slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        slots = 2 * cus;
    }
    return slots;
}

static int choose_splits(int64_t wgs, int n_db_tiles, int target_wgs) {
    if (wgs >= target_wgs) return 1;
    int64_t s = (target_wgs + wgs - 1) / wgs;
    const int64_t max_by_tiles = n_db_tiles / 64 > 0 ? n_db_tiles / 64 : 1;
    if (s > max_by_tiles) s = max_by_tiles;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return (int)s;
}

struct KnnPlan {
    int64_t main_wgs;   // workgroups of the un-split main launch (0 = none)
    int64_t tail_wgs;   // workgroups of the split launch (0 = none)
    int tail_splits;
};

static KnnPlan make_plan(int64_t nq, int n_db_tiles, int qb) {
    const int slots = device_slots() / qb;  // QB = 2 workgroups are alone on their CU
    const int64_t wgs = (nq + 128 * qb - 1) / (128 * qb);
    KnnPlan pl;
    if (wgs < 2 * (int64_t)slots) {  // small problem: one split launch
        pl.main_wgs = 0; pl.tail_wgs = wgs; pl.tail_splits = choose_splits(wgs, n_db_tiles, 2 * slots);
        return pl;
    }
    const int64_t rem = wgs % slots;
    if (rem == 0 || rem * 10 > (int64_t)slots * 9) {  // last round (almost) full: nothing to gain
        pl.main_wgs = wgs; pl.tail_wgs = 0; pl.tail_splits = 1;
        return pl;
    }
    pl.main_wgs = wgs - rem;
    pl.tail_wgs = rem;
    pl.tail_splits = choose_splits(rem, n_db_tiles, slots);
    return pl;
}

int64_t tdr_knn_workspace_bytes(int64_t nq, int64_t n_db, int k) {
    const int n_db_tiles = (int)((n_db + TILE_ROWS - 1) / TILE_ROWS);
    int64_t best = 0;  // upper bound over the workgroup shapes the launcher may pick
    for (int qb = 1; qb <= 2; ++qb) {
        const KnnPlan pl = make_plan(nq, n_db_tiles, qb);
        if (pl.tail_wgs == 0 || pl.tail_splits <= 1) continue;
        const int64_t tail_q = nq - pl.main_wgs * 128 * qb;
        const int64_t b = (int64_t)pl.tail_splits * tail_q * k * (int64_t)sizeof(uint64_t);
        if (b > best) best = b;
    }
    return best;
}

// Largest k the scan kernel supports for dimension d (LDS budget 160 KiB per workgroup).
int tdr_knn_max_k(int d) {
    const int kq = pick_kq(d);
    if (kq == 0) return 0;
    int k = 0;
    while (knn_lds_bytes(kq, k + 1, 1) <= 160 * 1024) ++k;
    return k;
}

/*
 * kNN of packed queries against a packed database.
 *   metric: 0 sqeuclidean, 1 euclidean, 2 angular.   exclude_self: skip database row q_offset + i.
 *   out_d (nq,k) fp32 ascending, out_i (nq,k) int32; rows ordered by (distance, index).
 * Requires 1 <= k <= min(tdr_knn_max_k(d), n_db - exclude_self).
 */
int tdr_knn_packed_f32(const float* qp, int64_t nq, int64_t q_offset, const float* yp, int64_t n_db, int d,
                       int k, int metric, int exclude_self, float* out_d, int32_t* out_i, void* ws,
                       int64_t ws_bytes, void* stream) {
    if (!qp || !yp || !out_d || !out_i || nq <= 0 || n_db <= 0 || d <= 0) return TDR_ERR_BAD_ARG;
    if (metric < 0 || metric > 2) return TDR_ERR_BAD_ARG;
    const int kq = pick_kq(d);
    if (kq == 0) return TDR_ERR_UNSUPPORTED;
    if (k < 1 || (int64_t)k > n_db - (exclude_self ? 1 : 0)) return TDR_ERR_BAD_ARG;
    if (k > tdr_knn_max_k(d)) return TDR_ERR_UNSUPPORTED;
    if (n_db > 0x7fffffffLL) return TDR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int n_db_tiles = (int)((n_db + TILE_ROWS - 1) / TILE_ROWS);
    const int qb = knn_qb(kq, k);
    const KnnPlan pl = make_plan(nq, n_db_tiles, qb);
    const size_t lds = knn_lds_bytes(kq, k, qb);
    const int64_t tile_f = tile_stride_floats(kq);
    const int qpw = 128 * qb;  // queries per workgroup
    // part 0 = main (un-split) launch, part 1 = tail launch (database sliced)
    for (int part = 0; part < 2; ++part) {
        const int64_t wgs = part == 0 ? pl.main_wgs : pl.tail_wgs;
        if (wgs == 0) continue;
        const int64_t q_begin = part == 0 ? 0 : pl.main_wgs * qpw;
        const int64_t q_count = part == 0 ? (pl.tail_wgs ? pl.main_wgs * qpw : nq) : nq - q_begin;
        KnnParams P;
        P.qp = qp + (q_begin / 32) * tile_f; P.yp = yp; P.nq = q_count; P.q_offset = q_offset + q_begin; P.n_db = n_db;
        P.k = k; P.metric = metric; P.exclude_self = exclude_self; P.n_db_tiles = n_db_tiles;
        P.n_splits = part == 0 ? 1 : pl.tail_splits;
        P.tiles_per_split = (P.n_db_tiles + P.n_splits - 1) / P.n_splits;
        P.out_d = out_d + q_begin * k; P.out_i = out_i + q_begin * k; P.ws_keys = (uint64_t*)ws;
        if (P.n_splits > 1) {
            const int64_t need = (int64_t)P.n_splits * q_count * k * (int64_t)sizeof(uint64_t);
            if (!ws || ws_bytes < need) return TDR_ERR_WORKSPACE;
        }
        int rc;
        switch (kq) {
            case 4: rc = launch_scan<4>(P, (int)wgs, lds, qb, st); break;
            case 8: rc = launch_scan<8>(P, (int)wgs, lds, qb, st); break;
            case 16: rc = launch_scan<16>(P, (int)wgs, lds, qb, st); break;
            default: rc = launch_scan<32>(P, (int)wgs, lds, qb, st); break;
        }
        if (rc != TDR_OK) return rc;
        if (P.n_splits > 1) {
            const size_t mlds = (size_t)4 * P.n_splits * k * sizeof(uint64_t);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_merge_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((q_count + 3) / 4)), dim3(256), mlds, st,
                               (const uint64_t*)ws, q_count, k, P.n_splits, metric, P.out_d, P.out_i);
            TDR_CHECK_LAUNCH();
        }
    }
    return TDR_OK;
}


/* ---- wide rows (D > 256): K-chunked scan, csrc/tdr_knn.hip knn_wide_kernel ----------------------------------------- */
static inline int wide_kq(int d) { return d > 256 ? ((d + 31) / 32) * 4 : 0; }
static size_t knn_wide_lds_bytes(int k) {
    return (size_t)2 * WIDE_TG * WIDE_KC * 256 * sizeof(float) + (size_t)2 * WIDE_TG * 64 * sizeof(float) +
           (size_t)4 * k * 32 * sizeof(uint64_t);
}

/* floats of the packed image of n rows of dimension d > 256 (0 otherwise) */
int64_t tdr_packed_floats_wide(int64_t n, int d) {
    const int kq = wide_kq(d);
    if (kq == 0 || n < 0) return 0;
    return ((n + TILE_ROWS - 1) / TILE_ROWS) * tile_stride_floats(kq);
}

int tdr_pack_rows_wide_f32(const float* X, int64_t n, int d, int64_t ldx, float* packed, float* norms_out, void* stream) {
    if (!X || !packed || n <= 0 || d <= 0 || ldx < d) return TDR_ERR_BAD_ARG;
    const int kq = wide_kq(d);
    if (kq == 0) return TDR_ERR_UNSUPPORTED;
    const int64_t tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
    hipLaunchKernelGGL(pack_wide_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, X, n, d, ldx, kq, packed, norms_out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* largest k of tdr_knn_wide_f32 (LDS budget of a workgroup) */
int tdr_knn_wide_max_k(void) {
    int k = 0;
    while (knn_wide_lds_bytes(k + 1) <= 160 * 1024) ++k;
    return k < 128 ? k : 128;
}

/* kNN of wide packed queries against a wide packed database (tdr_pack_rows_wide_f32 images); same contract as
 * tdr_knn_packed_f32 (workspace: tdr_knn_workspace_bytes). */
int tdr_knn_wide_f32(const float* qp, int64_t nq, int64_t q_offset, const float* yp, int64_t n_db, int d, int k, int metric,
                     int exclude_self, float* out_d, int32_t* out_i, void* ws, int64_t ws_bytes, void* stream) {
    if (!qp || !yp || !out_d || !out_i || nq <= 0 || n_db <= 0 || d <= 0) return TDR_ERR_BAD_ARG;
    if (metric < 0 || metric > 2) return TDR_ERR_BAD_ARG;
    const int kq = wide_kq(d);
    if (kq == 0) return TDR_ERR_UNSUPPORTED;
    if (k < 1 || (int64_t)k > n_db - (exclude_self ? 1 : 0)) return TDR_ERR_BAD_ARG;
    if (k > tdr_knn_wide_max_k() || n_db > 0x7fffffffLL) return TDR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int n_db_tiles = (int)((n_db + TILE_ROWS - 1) / TILE_ROWS);
    const KnnPlan pl = make_plan(nq, n_db_tiles, 1);
    const size_t lds = knn_wide_lds_bytes(k);
    const int64_t tile_f = tile_stride_floats(kq);
    for (int part = 0; part < 2; ++part) {
        const int64_t wgs = part == 0 ? pl.main_wgs : pl.tail_wgs;
        if (wgs == 0) continue;
        const int64_t q_begin = part == 0 ? 0 : pl.main_wgs * 128;
        const int64_t q_count = part == 0 ? (pl.tail_wgs ? pl.main_wgs * 128 : nq) : nq - q_begin;
        KnnParams P;
        P.qp = qp + (q_begin / 32) * tile_f; P.yp = yp; P.nq = q_count; P.q_offset = q_offset + q_begin; P.n_db = n_db;
        P.k = k; P.metric = metric; P.exclude_self = exclude_self; P.n_db_tiles = n_db_tiles; P.kq = kq;
        P.n_splits = part == 0 ? 1 : pl.tail_splits;
        P.tiles_per_split = (P.n_db_tiles + P.n_splits - 1) / P.n_splits;
        P.tiles_per_split = ((P.tiles_per_split + WIDE_TG - 1) / WIDE_TG) * WIDE_TG;  // whole groups per slice
        P.out_d = out_d + q_begin * k; P.out_i = out_i + q_begin * k; P.ws_keys = (uint64_t*)ws;
        if (P.n_splits > 1) {
            const int64_t need = (int64_t)P.n_splits * q_count * k * (int64_t)sizeof(uint64_t);
            if (!ws || ws_bytes < need) return TDR_ERR_WORKSPACE;
        }
        hipError_t e;
        if (k <= 64) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_wide_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(knn_wide_kernel<1>, dim3((unsigned)wgs, (unsigned)P.n_splits), dim3(256), lds, st, P);
        } else {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_wide_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(knn_wide_kernel<2>, dim3((unsigned)wgs, (unsigned)P.n_splits), dim3(256), lds, st, P);
        }
        TDR_CHECK_LAUNCH();
        if (P.n_splits > 1) {
            const size_t mlds = (size_t)4 * P.n_splits * k * sizeof(uint64_t);
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((q_count + 3) / 4)), dim3(256), mlds, st, (const uint64_t*)ws, q_count, k,
                               P.n_splits, metric, P.out_d, P.out_i);
            TDR_CHECK_LAUNCH();
        }
    }
    return TDR_OK;
}

/* Dense nq x n_db distance matrix from WIDE packed operands (D > 256; row stride ldo floats). */
int tdr_dense_dist_wide_f32(const float* qp, int64_t nq, int64_t q_offset, const float* yp, int64_t n_db, int d, int metric,
                            int exclude_self, float diag_add, float* out, int64_t ldo, void* stream) {
    if (!qp || !yp || !out || nq <= 0 || n_db <= 0 || ldo < n_db) return TDR_ERR_BAD_ARG;
    if (metric < 0 || metric > 2) return TDR_ERR_BAD_ARG;
    const int kq = wide_kq(d);
    if (kq == 0) return TDR_ERR_UNSUPPORTED;
    const unsigned gx = (unsigned)(((nq + 31) / 32 + 3) / 4);
    const unsigned gy = (unsigned)((n_db + 31) / 32);
    if (gy > 65535u) return TDR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dense_wide_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, qp, yp, nq, q_offset, n_db, kq, metric,
                       exclude_self, diag_add, out, ldo);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Dense nq x n_db distance matrix from packed operands (row stride ldo floats). */
int tdr_dense_dist_packed_f32(const float* qp, int64_t nq, int64_t q_offset, const float* yp, int64_t n_db,
                              int d, int metric, int exclude_self, float diag_add, float* out, int64_t ldo,
                              void* stream) {
    if (!qp || !yp || !out || nq <= 0 || n_db <= 0 || ldo < n_db) return TDR_ERR_BAD_ARG;
    if (metric < 0 || metric > 2) return TDR_ERR_BAD_ARG;
    const int kq = pick_kq(d);
    if (kq == 0) return TDR_ERR_UNSUPPORTED;
    const int64_t n_qtiles = (nq + 31) / 32;
    const unsigned gx = (unsigned)((n_qtiles + 3) / 4);
    const unsigned gy = (unsigned)((n_db + 31) / 32);
    if (gy > 65535u) return TDR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    switch (kq) {
        case 4: hipLaunchKernelGGL(dense_dist_kernel<4>, dim3(gx, gy), dim3(256), 0, st, qp, yp, nq, q_offset, n_db, metric, exclude_self, diag_add, out, ldo); break;
        case 8: hipLaunchKernelGGL(dense_dist_kernel<8>, dim3(gx, gy), dim3(256), 0, st, qp, yp, nq, q_offset, n_db, metric, exclude_self, diag_add, out, ldo); break;
        case 16: hipLaunchKernelGGL(dense_dist_kernel<16>, dim3(gx, gy), dim3(256), 0, st, qp, yp, nq, q_offset, n_db, metric, exclude_self, diag_add, out, ldo); break;
        default: hipLaunchKernelGGL(dense_dist_kernel<32>, dim3(gx, gy), dim3(256), 0, st, qp, yp, nq, q_offset, n_db, metric, exclude_self, diag_add, out, ldo); break;
    }
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Fraction of the K indices of row i of a that also occur in row i of b (both (n, K) int32). */
int tdr_knn_overlap_i32(const int32_t* a, const int32_t* b, int64_t n, int K, float* out, void* stream) {
    if (!a || !b || !out || n <= 0 || K <= 0) return TDR_ERR_BAD_ARG;
    if (K > 2048) return TDR_ERR_UNSUPPORTED;
    const size_t lds = (size_t)4 * K * sizeof(int32_t);
    hipLaunchKernelGGL(knn_overlap_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), lds, (hipStream_t)stream, a, b, n, K, out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* largest k of the running top-k lists (tdr_topk_merge_f32 / tdr_topk_merge_cand_f32) */
int tdr_topk_max_k(void) { return TOPK_MERGE_MAX_K; }

/* General-D kNN, step 1: run_keys (nq, k) <- empty lists. */
int tdr_topk_init(uint64_t* run_keys, int64_t nq, int k, void* stream) {
    if (!run_keys || nq <= 0 || k <= 0) return TDR_ERR_BAD_ARG;
    const int64_t total = nq * k;
    hipLaunchKernelGGL(fill_keys_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, run_keys, total);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Step 2 (per database chunk): fold G = Xq Yc^T (nq x nd, row stride ldg) into the running lists (metric 3, manhattan:
 * G already holds the distances, tdr_l1_block_f32).  xn / yn: squared norms of the queries / of this chunk; q_global0 / d_global0: global indices of query 0 and of chunk row 0. */
int tdr_topk_merge_f32(const float* G, int64_t ldg, int64_t nq, int64_t nd, const float* xn, const float* yn,
                       int64_t q_global0, int64_t d_global0, int k, int metric, int exclude_self, uint64_t* run_keys,
                       void* stream) {
    if (!G || !run_keys || nq <= 0 || nd <= 0 || ldg < nd || k <= 0) return TDR_ERR_BAD_ARG;
    if (metric < 0 || metric > 4 || ((metric < 2 || metric == 4) && (!xn || !yn))) return TDR_ERR_BAD_ARG;
    if (k > TOPK_MERGE_MAX_K) return TDR_ERR_UNSUPPORTED;
    TopkMergeParams P;
    P.G = G; P.ldg = ldg; P.nq = nq; P.nd = nd; P.xn = xn; P.yn = yn; P.q_global0 = q_global0; P.d_global0 = d_global0;
    P.k = k; P.metric = metric; P.exclude_self = exclude_self; P.run_keys = run_keys; P.cand = nullptr;
    return launch_topk_merge(P, stream);
}

/* Same fold for per-query candidate lists: E (nq, nc) distances and cand (nq, nc) database indices (both row stride
 * ld; negative indices are skipped).  Used to rank exactly re-evaluated candidates by (distance, index). */
int tdr_topk_merge_cand_f32(const float* E, const int32_t* cand, int64_t ld, int64_t nq, int64_t nc, int k,
                            uint64_t* run_keys, void* stream) {
    if (!E || !cand || !run_keys || nq <= 0 || nc <= 0 || ld < nc || k <= 0) return TDR_ERR_BAD_ARG;
    if (k > TOPK_MERGE_MAX_K) return TDR_ERR_UNSUPPORTED;
    TopkMergeParams P;
    P.G = E; P.ldg = ld; P.nq = nq; P.nd = nc; P.xn = nullptr; P.yn = nullptr; P.q_global0 = 0; P.d_global0 = 0;
    P.k = k; P.metric = 3; P.exclude_self = 0; P.run_keys = run_keys; P.cand = cand;
    return launch_topk_merge(P, stream);
}

/* metric "sqhyperbolic" (distance/torch.py:101-107), dense form: G = X Y^T (nq x nd, row stride ld) is overwritten with
 * arccosh(1 + 2 relu(|x|^2 + |y|^2 - 2 G) / ((1 - |x|^2)(1 - |y|^2)) + 1e-8)^2.  (kNN: tdr_topk_merge_f32, metric 4.) */
int tdr_hyperbolic_from_gram_f32(float* G, int64_t ld, int64_t nq, int64_t nd, const float* xn, const float* yn,
                                 void* stream) {
    if (!G || !xn || !yn || nq < 0 || nd < 0 || ld < nd) return TDR_ERR_BAD_ARG;
    if (nq == 0 || nd == 0) return TDR_OK;
    if (nq > 65535) return TDR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(hyperbolic_epilogue_kernel, dim3((unsigned)((nd + 255) / 256), (unsigned)nq), dim3(256), 0,
                       (hipStream_t)stream, G, ld, nq, nd, xn, yn);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Step 3: lists -> out_d (nq, k) fp32 ascending, out_i (nq, k) int32. */
int tdr_topk_emit_f32(const uint64_t* run_keys, int64_t nq, int k, int metric, float* out_d, int32_t* out_i, void* stream) {
    if (!run_keys || !out_d || !out_i || nq <= 0 || k <= 0) return TDR_ERR_BAD_ARG;
    const int64_t total = nq * k;
    hipLaunchKernelGGL(topk_emit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, run_keys,
                       total, metric, out_d, out_i);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
