// K1m -- Manhattan (L1) pairwise distances.
//
// Replaces (citations under /root/reference/torchdr):
//   distance/torch.py:96-98     C = (X.unsqueeze(-2) - Y.unsqueeze(-3)).abs().sum(-1)   (an (n, m, d) intermediate there)
//   distance/base.py:368, 388   torch.cdist(p=1) / the gathered form (the latter lives in tdr_affinity.hip)
//
// |x - y| does not factor into a contraction, so this is VALU work, not MFMA work: 1.5 instructions per (pair, feature)
// (a packed subtract for two columns, then v_add with the |.| source modifier).  A 256-thread workgroup owns a 128 x 128 tile of the output; each thread accumulates an 8 x 8
// sub-tile in registers (128 VALU instructions per 4 LDS reads of 16 B), the operands travel through LDS in
// feature-chunks of 16, transposed so that a thread's 8 rows / 8 columns are two ds_read_b128 each.  The host feeds
// the block to tdr_topk_merge_f32 (metric 3) for kNN, or keeps it as the dense matrix.
#include "tdr_common.h"

namespace tdr {

constexpr int L1_T = 128;    // tile edge (queries and database rows)
constexpr int L1_DC = 16;    // features per LDS stage
constexpr int L1_LD = L1_T + 4;
typedef float v2f __attribute__((ext_vector_type(2)));

struct L1Params {
    const float* X; int64_t ldx, nq;
    const float* Y; int64_t ldy, nd;
    int d;
    float* out; int64_t ldo;
};

// 8 consecutive features of one row starting at c0 (zero beyond d)
__device__ __forceinline__ void l1_load8(const float* __restrict__ row, int c0, int d, bool vec, float (&v)[8]) {
    if (vec && c0 + 8 <= d) {
        const float4 a = *reinterpret_cast<const float4*>(row + c0);
        const float4 b = *reinterpret_cast<const float4*>(row + c0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (c0 + i < d) ? row[c0 + i] : 0.f;
    }
}

__global__ __launch_bounds__(256) void l1_block_kernel(const L1Params P) {
    __shared__ __attribute__((aligned(16))) float Xs[L1_DC][L1_LD];
    __shared__ __attribute__((aligned(16))) float Ys[L1_DC][L1_LD];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;                 // 16 x 16 threads, 8 x 8 outputs each
    const int64_t q0 = (int64_t)blockIdx.y * L1_T, j0 = (int64_t)blockIdx.x * L1_T;
    // staging role: row (tid >> 1) of the tile, features [half*8, half*8+8) of the chunk
    const int srow = tid >> 1, half = tid & 1;
    const int64_t xr = q0 + srow < P.nq ? q0 + srow : P.nq - 1;
    const int64_t yr = j0 + srow < P.nd ? j0 + srow : P.nd - 1;
    const float* xrow = P.X + (size_t)xr * P.ldx;
    const float* yrow = P.Y + (size_t)yr * P.ldy;
    const bool vec = (P.ldx % 4 == 0) && (P.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(P.X) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(P.Y) & 15) == 0);

    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;

    float xv[8], yv[8];
    l1_load8(xrow, half * 8, P.d, vec, xv);
    l1_load8(yrow, half * 8, P.d, vec, yv);
    for (int c0 = 0; c0 < P.d; c0 += L1_DC) {
        __syncthreads();                                    // previous chunk fully consumed
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            Xs[half * 8 + i][srow] = xv[i];
            Ys[half * 8 + i][srow] = yv[i];
        }
        __syncthreads();
        if (c0 + L1_DC < P.d) {                             // next chunk's global loads fly under this chunk's math
            l1_load8(xrow, c0 + L1_DC + half * 8, P.d, vec, xv);
            l1_load8(yrow, c0 + L1_DC + half * 8, P.d, vec, yv);
        }
#pragma unroll 2
        for (int dd = 0; dd < L1_DC; ++dd) {
            const float4 a0 = *reinterpret_cast<const float4*>(&Xs[dd][ty * 8]);
            const float4 a1 = *reinterpret_cast<const float4*>(&Xs[dd][ty * 8 + 4]);
            const float4 b0 = *reinterpret_cast<const float4*>(&Ys[dd][tx * 8]);
            const float4 b1 = *reinterpret_cast<const float4*>(&Ys[dd][tx * 8 + 4]);
            const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const v2f b[4] = {{b0.x, b0.y}, {b0.z, b0.w}, {b1.x, b1.y}, {b1.z, b1.w}};
            // 1.5 instructions per (pair, feature): one packed subtract per two columns, then v_add with the |.| source
            // modifier (written as asm so that the accumulations are not re-packed behind a v_and)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v2f t = (v2f){a[i], a[i]} - b[j];
                    asm("v_add_f32 %0, %0, |%1|" : "+v"(acc[i][2 * j]) : "v"(t.x));
                    asm("v_add_f32 %0, %0, |%1|" : "+v"(acc[i][2 * j + 1]) : "v"(t.y));
                }
        }
    }

    const int64_t jb = j0 + tx * 8;
    const bool vst = (P.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(P.out) & 15) == 0) && jb + 8 <= P.nd;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t qi = q0 + ty * 8 + i;
        if (qi >= P.nq) break;
        float* o = P.out + (size_t)qi * P.ldo + jb;
        if (vst) {
            *reinterpret_cast<float4*>(o) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(acc[i][4], acc[i][5], acc[i][6], acc[i][7]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (jb + j < P.nd) o[j] = acc[i][j];
        }
    }
}

// ---- exact re-evaluation in the reference's summation order ------------------------------------------------------
// The CPU reference reduces the materialised |x - y| row with ATen's vectorised inner sum (AVX2 build): 8 lanes x 4
// interleaved rows = 32 running sums (element k feeds sum k mod 32), dumped into a second level every 16 rounds, then
// leftover 8-wide items join lanes of row 0, rows 1..3 fold into row 0, the scalar tail and the 8 lanes are added in
// order (oracle/knn_oracle.c restates it and is pinned bit for bit against the reference).  One thread per (query,
// candidate) pair walks exactly that order, so the value it returns IS the reference's float.  Valid for d < 8192
// (beyond, a third cascade level would start).
constexpr int L1_EXACT_MAX_D = 8192;

struct L1ExactParams {
    const float* X; int64_t ldx;
    const int64_t* q_rows;       // nullptr: query i is row i of X
    int64_t nq, q_global0;       // global id of query i = q_global0 + its row of X (self exclusion)
    const float* Y; int64_t ldy;
    const int32_t* cand;         // (nq, ldc) database rows per query, or nullptr: dense columns [j0, j0 + nc)
    int64_t ldc, nc, j0;
    int d, exclude_self;
    float* out; int64_t ldo;
};

template <bool VEC>
__device__ __forceinline__ void l1_terms32(const float* __restrict__ x, const float* __restrict__ y, int k0, float (&a)[32]) {
    if (VEC) {
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(x + k0 + c);
            const float4 yv = *reinterpret_cast<const float4*>(y + k0 + c);
            a[c] = __fadd_rn(a[c], fabsf(__fsub_rn(xv.x, yv.x)));
            a[c + 1] = __fadd_rn(a[c + 1], fabsf(__fsub_rn(xv.y, yv.y)));
            a[c + 2] = __fadd_rn(a[c + 2], fabsf(__fsub_rn(xv.z, yv.z)));
            a[c + 3] = __fadd_rn(a[c + 3], fabsf(__fsub_rn(xv.w, yv.w)));
        }
    } else {
#pragma unroll
        for (int c = 0; c < 32; ++c) a[c] = __fadd_rn(a[c], fabsf(__fsub_rn(x[k0 + c], y[k0 + c])));
    }
}

template <bool VEC>
__device__ float l1_exact_sum(const float* __restrict__ x, const float* __restrict__ y, int d) {
    auto term = [&](int k) { return fabsf(__fsub_rn(x[k], y[k])); };
    if (d < 8) {  // scalar path: 4 interleaved rows of width 1
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        const int size_ilp = d / 4;
        if (size_ilp)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[r] = __fadd_rn(part[r], term(r));
        for (int i = size_ilp * 4; i < d; ++i) part[0] = __fadd_rn(part[0], term(i));
#pragma unroll
        for (int r = 1; r < 4; ++r) part[0] = __fadd_rn(part[0], part[r]);
        return part[0];
    }
    const int vec_size = d / 8, size_ilp = vec_size / 4;
    float a0[32], a1[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { a0[c] = 0.f; a1[c] = 0.f; }
    int i = 0;
    for (; i + 16 <= size_ilp;) {
        for (int j = 0; j < 16; ++j, ++i) l1_terms32<VEC>(x, y, i * 32, a0);
#pragma unroll
        for (int c = 0; c < 32; ++c) { a1[c] = __fadd_rn(a1[c], a0[c]); a0[c] = 0.f; }
    }
    for (; i < size_ilp; ++i) l1_terms32<VEC>(x, y, i * 32, a0);
#pragma unroll
    for (int c = 0; c < 32; ++c) a0[c] = __fadd_rn(a0[c], a1[c]);
    for (int it = size_ilp * 4; it < vec_size; ++it)       // leftover 8-wide items join row 0
#pragma unroll
        for (int l = 0; l < 8; ++l) a0[l] = __fadd_rn(a0[l], term(it * 8 + l));
#pragma unroll
    for (int r = 1; r < 4; ++r)
#pragma unroll
        for (int l = 0; l < 8; ++l) a0[l] = __fadd_rn(a0[l], a0[r * 8 + l]);
    float fin = 0.f;
    for (int k = vec_size * 8; k < d; ++k) fin = __fadd_rn(fin, term(k));
#pragma unroll
    for (int l = 0; l < 8; ++l) fin = __fadd_rn(fin, a0[l]);
    return fin;
}

__global__ __launch_bounds__(256) void l1_exact_kernel(const L1ExactParams P, int vec) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P.nq * P.nc) return;
    const int64_t i = idx / P.nc, c = idx - i * P.nc;
    const int64_t qrow = P.q_rows ? P.q_rows[i] : i;
    const int64_t qid = P.q_global0 + qrow;
    int64_t j;
    if (P.cand) {
        j = P.cand[(size_t)i * P.ldc + c];
        if (j < 0) { P.out[(size_t)i * P.ldo + c] = __builtin_inff(); return; }
    } else {
        j = P.j0 + c;
    }
    float v;
    if (P.exclude_self && j == qid) v = __builtin_inff();
    else {
        const float* x = P.X + (size_t)qrow * P.ldx;
        const float* y = P.Y + (size_t)j * P.ldy;
        v = vec ? l1_exact_sum<true>(x, y, P.d) : l1_exact_sum<false>(x, y, P.d);
    }
    P.out[(size_t)i * P.ldo + c] = v;
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* out[i][j] = sum_c |X[i][c] - Y[j][c]| for an (nq x nd) block; rows of X / Y / out have strides ldx / ldy / ldo. */
int tdr_l1_block_f32(const float* X, int64_t ldx, int64_t nq, const float* Y, int64_t ldy, int64_t nd, int d, float* out,
                     int64_t ldo, void* stream) {
    if (!X || !Y || !out || nq < 0 || nd < 0 || d <= 0 || ldx < d || ldy < d || ldo < nd) return TDR_ERR_BAD_ARG;
    if (nq == 0 || nd == 0) return TDR_OK;
    const int64_t gx = (nd + L1_T - 1) / L1_T, gy = (nq + L1_T - 1) / L1_T;
    if (gy > 65535) return TDR_ERR_UNSUPPORTED;            // the host walks the queries in blocks far below this
    L1Params P;
    P.X = X; P.ldx = ldx; P.nq = nq; P.Y = Y; P.ldy = ldy; P.nd = nd; P.d = d; P.out = out; P.ldo = ldo;
    hipLaunchKernelGGL(l1_block_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, P);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Manhattan distances in the reference's exact summation order (see l1_exact_sum): out[i][c] for query i (row
 * q_rows[i] of X, or row i when q_rows is NULL; global id = q_global0 + that row) against database row
 * cand[i][c] (cand != NULL; a negative entry gives +inf) or j0 + c (cand == NULL), c < nc.  exclude_self: the
 * query's own database row gives +inf.  d < 8192. */
int tdr_l1_exact_f32(const float* X, int64_t ldx, const int64_t* q_rows, int64_t nq, int64_t q_global0, const float* Y,
                     int64_t ldy, const int32_t* cand, int64_t ldc, int64_t nc, int64_t j0, int d, int exclude_self,
                     float* out, int64_t ldo, void* stream) {
    if (!X || !Y || !out || nq < 0 || nc < 0 || d <= 0 || ldx < d || ldy < d || ldo < nc) return TDR_ERR_BAD_ARG;
    if (cand && ldc < nc) return TDR_ERR_BAD_ARG;
    if (d >= L1_EXACT_MAX_D) return TDR_ERR_UNSUPPORTED;
    if (nq == 0 || nc == 0) return TDR_OK;
    L1ExactParams P;
    P.X = X; P.ldx = ldx; P.q_rows = q_rows; P.nq = nq; P.q_global0 = q_global0; P.Y = Y; P.ldy = ldy; P.cand = cand;
    P.ldc = ldc; P.nc = nc; P.j0 = j0; P.d = d; P.exclude_self = exclude_self; P.out = out; P.ldo = ldo;
    const int vec = (ldx % 4 == 0) && (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
    const int64_t total = nq * nc;
    if ((total + 255) / 256 > 0x7fffffffLL) return TDR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(l1_exact_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, vec);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
