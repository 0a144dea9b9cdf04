// float64 twins of the all-pairs reductions behind SymmetricEntropicAffinity, the Student-kernel Sinkhorn updates and the
// TSNEkhorn forces (tdr_dense.hip holds the float32 forms: matrix-free fp32-MFMA pair scans).  The reference computes in the
// dtype of its input (tests/test_neighbor_embedding.py:34,55-74 run every method in float64 too), so a float64 block is taken
// through SNEkhorn in float64:
//
//   affinity/entropic.py:518-534        row statistics of exp((mu_i + mu_j - 2 C_ij) / (e_i + e_j))   -> tdr_sea_rowstats_dense_f64
//   affinity/entropic.py:728-748        s_j = sum_i v_i / (1 + |z_i - z_j|^2) (Sinkhorn update, adjoint) -> tdr_student_sum_f64
//   neighbor_embedding/tsnekhorn.py:210-230   force 4 sum_i (P_ij - Q_ij) w_ij (z_j - z_i)            -> tdr_khorn_grad_dense_f64
//   neighbor_embedding/tsnekhorn.py:134,224-227  the same through the unrolled Sinkhorn updates      -> tdr_khorn_grad_unrolled_dense_f64
//
// These are parity runs, not the fast path: the input-side kernels read a DENSE float64 matrix of squared distances (what the
// reference itself materialises; tdr_knn_f64 with k = 0 writes it on the fp64 matrix pipe, bitwise symmetric), so N is bounded by
// that matrix (the host caps it at 16384 points = 2 GiB and falls back to the float32 path beyond).  One thread owns one row j
// and walks the columns i in order; the column's per-point values are wavefront-uniform loads, C[i][j] is read along j
// (coalesced; C is symmetric).  Columns are cut into segments over blockIdx.y so that a few thousand rows still fill the
// chip; every segment writes its partial sums to its own plane and the planes are added in segment order: deterministic.
// Every state is a plain sum (no running maximum: exp() of a float64 log-affinity does not overflow where the reference's own
// `P.exp()` does not).
#include "tdr_common.h"

namespace tdr {

struct Pairs64Params {
    const double* C; int64_t ldc;   // dense squared distances (NULL for the embedding-only sums)
    const double* side; int side_w; // (n, side_w) per-point values
    int64_t n, cols_per_seg;
    double c0, c1;
    int zero_diag; double diag_add;
    double* planes;                 // (n_seg, n, NS)
};

// S = sum_i p, T = sum_i p lp, U = sum_i p C_ij with lp = (mu_j + mu_i - 2 C_ij) / (e_j + e_i), p = exp(lp); side = (mu, e)
struct SeaStats64 {
    static constexpr int NS = 3;
    static constexpr bool NEEDS_C = true;
    double mu, e, s, t, u;
    __device__ __forceinline__ void init(const double* sj, const Pairs64Params&) { mu = sj[0]; e = sj[1]; s = 0.0; t = 0.0; u = 0.0; }
    __device__ __forceinline__ void add(double c, const double* si, bool, const Pairs64Params&) {
        const double lp = (mu + si[0] - 2.0 * c) / (e + si[1]);
        const double p = exp(lp);
        s += p;
        t = fma(p, lp, t);
        u = p > 0.0 ? fma(p, c, u) : u;     // an excluded diagonal carries c = 1e12 and p = 0
    }
    __device__ __forceinline__ void save(double* o) const { o[0] = s; o[1] = t; o[2] = u; }
};

// s_j = sum_i v_i / (1 + |z_j - z_i|^2 [+ diag_add on the diagonal]); side = (z[NC], v)
template <int NC>
struct StudentSum64 {
    static constexpr int NS = 1;
    static constexpr bool NEEDS_C = false;
    double z[NC], s;
    __device__ __forceinline__ void init(const double* sj, const Pairs64Params&) {
#pragma unroll
        for (int c = 0; c < NC; ++c) z[c] = sj[c];
        s = 0.0;
    }
    __device__ __forceinline__ void add(double, const double* si, bool diag, const Pairs64Params& P) {
        double d = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { const double df = z[c] - si[c]; d = fma(df, df, d); }
        if (diag && P.zero_diag) d += P.diag_add;
        s = fma(si[NC], 1.0 / (1.0 + d), s);
    }
    __device__ __forceinline__ void save(double* o) const { o[0] = s; }
};

// g_j = sum_i (P_ij - Q_ij) w_ij (z_j - z_i), P_ij = exp(lp_ij - log N), Q_ij = E_j E_i w_ij / N; side = (mu, e, z[NC], E);
// c0 = log N, c1 = 1 / N (the caller multiplies by 4)
template <int NC>
struct KhornForce64 {
    static constexpr int NS = NC;
    static constexpr bool NEEDS_C = true;
    double mu, e, z[NC], E, g[NC];
    __device__ __forceinline__ void init(const double* sj, const Pairs64Params&) {
        mu = sj[0]; e = sj[1]; E = sj[2 + NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { z[c] = sj[2 + c]; g[c] = 0.0; }
    }
    __device__ __forceinline__ void add(double c, const double* si, bool, const Pairs64Params& P) {
        const double p = exp((mu + si[0] - 2.0 * c) / (e + si[1]) - P.c0);
        double df[NC], d2 = 1.0;
#pragma unroll
        for (int k = 0; k < NC; ++k) { df[k] = z[k] - si[2 + k]; d2 = fma(df[k], df[k], d2); }
        const double w = 1.0 / d2;
        const double q = E * si[2 + NC] * w * P.c1;
        const double coef = (p - q) * w;
#pragma unroll
        for (int k = 0; k < NC; ++k) g[k] = fma(coef, df[k], g[k]);
    }
    __device__ __forceinline__ void save(double* o) const {
#pragma unroll
        for (int k = 0; k < NC; ++k) o[k] = g[k];
    }
};

// the force through the K <= 5 unrolled Sinkhorn updates (tdr_dense.hip: KhornForceUnrolled): the Q term replaced by the
// bilinear form w_ij sum_k (a_jk b_ik + a_ik b_jk); side = (mu, e, z[NC], a[5], b[5])
template <int NC>
struct KhornForceUnrolled64 {
    static constexpr int KU = 5;
    static constexpr int NS = NC;
    static constexpr bool NEEDS_C = true;
    double mu, e, z[NC], a[KU], b[KU], g[NC];
    __device__ __forceinline__ void init(const double* sj, const Pairs64Params&) {
        mu = sj[0]; e = sj[1];
#pragma unroll
        for (int c = 0; c < NC; ++c) { z[c] = sj[2 + c]; g[c] = 0.0; }
#pragma unroll
        for (int k = 0; k < KU; ++k) { a[k] = sj[2 + NC + k]; b[k] = sj[2 + NC + KU + k]; }
    }
    __device__ __forceinline__ void add(double c, const double* si, bool, const Pairs64Params& P) {
        const double p = exp((mu + si[0] - 2.0 * c) / (e + si[1]) - P.c0);
        double df[NC], d2 = 1.0;
#pragma unroll
        for (int k = 0; k < NC; ++k) { df[k] = z[k] - si[2 + k]; d2 = fma(df[k], df[k], d2); }
        const double w = 1.0 / d2;
        double bil = 0.0;
#pragma unroll
        for (int k = 0; k < KU; ++k) bil = fma(a[k], si[2 + NC + KU + k], fma(si[2 + NC + k], b[k], bil));
        const double coef = fma(w, bil, p) * w;
#pragma unroll
        for (int k = 0; k < NC; ++k) g[k] = fma(coef, df[k], g[k]);
    }
    __device__ __forceinline__ void save(double* o) const {
#pragma unroll
        for (int k = 0; k < NC; ++k) o[k] = g[k];
    }
};

template <class Epi>
__global__ __launch_bounds__(64) void pairs64_kernel(const Pairs64Params P) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool have = j < P.n;
    const int64_t jj = have ? j : P.n - 1;      // idle lanes shadow the last row (loads stay in bounds, nothing is stored)
    Epi epi;
    epi.init(P.side + (size_t)jj * P.side_w, P);
    const int64_t i_lo = (int64_t)blockIdx.y * P.cols_per_seg;
    const int64_t i_hi = i_lo + P.cols_per_seg < P.n ? i_lo + P.cols_per_seg : P.n;
    for (int64_t i = i_lo; i < i_hi; ++i) {
        const double c = Epi::NEEDS_C ? P.C[(size_t)i * P.ldc + jj] : 0.0;
        epi.add(c, P.side + (size_t)i * P.side_w, i == jj, P);
    }
    if (have) epi.save(P.planes + ((size_t)blockIdx.y * P.n + j) * Epi::NS);
}

__global__ __launch_bounds__(256) void sum_planes64_kernel(const double* __restrict__ planes, int n_planes, int64_t cnt, double scale,
                                                           double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt) return;
    double a = planes[i];
    for (int p = 1; p < n_planes; ++p) a += planes[(size_t)p * cnt + i];
    out[i] = scale * a;
}

static int pairs64_segments(int64_t n) {
    const int64_t blocks = (n + 63) / 64;
    int64_t seg = (4096 + blocks - 1) / blocks;    // ~4 wavefronts per SIMD over the chip
    if (seg < 1) seg = 1;
    if (seg > 64) seg = 64;
    if (seg > n) seg = n;
    return (int)seg;
}

template <class Epi>
static int run_pairs64(Pairs64Params P, double scale, double* out, void* ws, int64_t ws_bytes, hipStream_t st) {
    const int seg = pairs64_segments(P.n);
    const int64_t need = (int64_t)seg * P.n * Epi::NS * (int64_t)sizeof(double);
    if (!ws || ws_bytes < need) return TDR_ERR_WORKSPACE;
    P.planes = (double*)ws;
    P.cols_per_seg = (P.n + seg - 1) / seg;
    const dim3 grid((unsigned)((P.n + 63) / 64), (unsigned)seg);
    hipLaunchKernelGGL((pairs64_kernel<Epi>), grid, dim3(64), 0, st, P);
    const int64_t cnt = P.n * Epi::NS;
    hipLaunchKernelGGL(sum_planes64_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, (const double*)P.planes, seg, cnt, scale, out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* largest row count the dense float64 forms take (the N x N float64 distance matrix is 2 GiB there) */
int64_t tdr_pairs_f64_max_rows(void) { return 16384; }

/* bytes of the plane workspace of the four entry points below for n rows and n_state sums per row */
int64_t tdr_pairs_f64_workspace_bytes(int64_t n, int n_state) {
    if (n <= 0 || n_state <= 0) return 0;
    return (int64_t)pairs64_segments(n) * n * n_state * (int64_t)sizeof(double);
}

/* affinity/entropic.py:518-534 in float64 on the dense matrix C (n, n; row stride ldc; symmetric -- tdr_knn_f64 with k = 0;
 * an excluded diagonal holds + 1e12): side (n, 2) = (mu, e); out (n, 3) = (sum_i p_ij, sum_i p_ij lp_ij, sum_i p_ij C_ij)
 * with lp_ij = (mu_i + mu_j - 2 C_ij) / (e_i + e_j), p = exp(lp).  P_sum = out[:, 0], H = -(out[:, 1] - out[:, 0]). */
int tdr_sea_rowstats_dense_f64(const double* C, int64_t n, int64_t ldc, const double* side, double* out, void* ws, int64_t ws_bytes,
                               void* stream) {
    if (!C || !side || !out || n <= 0 || ldc < n) return TDR_ERR_BAD_ARG;
    Pairs64Params P = {};
    P.C = C; P.ldc = ldc; P.side = side; P.side_w = 2; P.n = n;
    return run_pairs64<SeaStats64>(P, 1.0, out, ws, ws_bytes, (hipStream_t)stream);
}

/* out_j = sum_i v_i / (1 + |z_j - z_i|^2), the diagonal term with + diag_add in the denominator when zero_diag
 * (affinity/entropic.py:728-748 with the Student base kernel; tdr_sinkhorn_pass_f32 / tdr_student_matvec_f32 in float64).
 * side (n, nc + 1) = (z, v); nc = 2, 3 or 4. */
int tdr_student_sum_f64(const double* side, int nc, int64_t n, int zero_diag, double diag_add, double* out, void* ws, int64_t ws_bytes,
                        void* stream) {
    if (!side || !out || n <= 0) return TDR_ERR_BAD_ARG;
    if (nc < 2 || nc > 4) return TDR_ERR_UNSUPPORTED;
    Pairs64Params P = {};
    P.side = side; P.side_w = nc + 1; P.n = n; P.zero_diag = zero_diag; P.diag_add = diag_add;
    hipStream_t st = (hipStream_t)stream;
    if (nc == 2) return run_pairs64<StudentSum64<2>>(P, 1.0, out, ws, ws_bytes, st);
    if (nc == 3) return run_pairs64<StudentSum64<3>>(P, 1.0, out, ws, ws_bytes, st);
    return run_pairs64<StudentSum64<4>>(P, 1.0, out, ws, ws_bytes, st);
}

/* tsnekhorn.py:210-230 in float64: grad (n, nc) = 4 sum_i (P_ij - Q_ij) w_ij (z_j - z_i); side (n, nc + 3) = (mu, e, z, E = exp(dual)),
 * P_ij = exp(lp_ij - log_n), Q_ij = E_i E_j w_ij / n. */
int tdr_khorn_grad_dense_f64(const double* C, int64_t n, int64_t ldc, const double* side, int nc, double log_n, double* grad, void* ws,
                             int64_t ws_bytes, void* stream) {
    if (!C || !side || !grad || n <= 0 || ldc < n) return TDR_ERR_BAD_ARG;
    if (nc < 2 || nc > 4) return TDR_ERR_UNSUPPORTED;
    Pairs64Params P = {};
    P.C = C; P.ldc = ldc; P.side = side; P.side_w = nc + 3; P.n = n; P.c0 = log_n; P.c1 = 1.0 / (double)n;
    hipStream_t st = (hipStream_t)stream;
    if (nc == 2) return run_pairs64<KhornForce64<2>>(P, 4.0, grad, ws, ws_bytes, st);
    if (nc == 3) return run_pairs64<KhornForce64<3>>(P, 4.0, grad, ws, ws_bytes, st);
    return run_pairs64<KhornForce64<4>>(P, 4.0, grad, ws, ws_bytes, st);
}

/* tsnekhorn.py:134,224-227 in float64 (tdr_khorn_grad_unrolled_f32's form): side (n, nc + 12) = (mu, e, z, a[5], b[5]). */
int tdr_khorn_grad_unrolled_dense_f64(const double* C, int64_t n, int64_t ldc, const double* side, int nc, double log_n, double* grad,
                                      void* ws, int64_t ws_bytes, void* stream) {
    if (!C || !side || !grad || n <= 0 || ldc < n) return TDR_ERR_BAD_ARG;
    if (nc < 2 || nc > 4) return TDR_ERR_UNSUPPORTED;
    Pairs64Params P = {};
    P.C = C; P.ldc = ldc; P.side = side; P.side_w = nc + 12; P.n = n; P.c0 = log_n;
    hipStream_t st = (hipStream_t)stream;
    if (nc == 2) return run_pairs64<KhornForceUnrolled64<2>>(P, 4.0, grad, ws, ws_bytes, st);
    if (nc == 3) return run_pairs64<KhornForceUnrolled64<3>>(P, 4.0, grad, ws, ws_bytes, st);
    return run_pairs64<KhornForceUnrolled64<4>>(P, 4.0, grad, ws, ws_bytes, st);
}

}  // extern "C"
