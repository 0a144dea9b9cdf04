// float64 twins of the affinity side of the path (K1 - K4 + the gathered distances): the reference computes in the
// dtype of its input (its tests run every affinity in float32 AND float64, tests/test_affinity.py:54-60), so float64
// inputs get float64 arithmetic here too.  Same reference lines as the float32 kernels:
//   distance/torch.py:21-125 + utils/utils.py:173-216   pairwise distances / kmin        -> knn_f64_kernel
//   utils/root_search.py + affinity/knn_normalized.py:445-465, entropic.py:272-310      -> *_search_f64_kernel
//   utils/sparse.py:7-206 (values of P + P^T - P o P^T on the float32 pipeline's pattern) -> sym_values_f64_kernel
//   distance/base.py:384-398                                                              -> indexed_sqdist_f64_kernel
// The contraction runs on the fp64 matrix pipe (v_mfma_f64_16x16x4_f64, 78.6 TFLOP/s peak).  Bit parity with MKL's
// dgemm is not attempted (its blocking is not a single k-ordered chain): tests hold these kernels to 1e-12 relative and
// to identical neighbour sets wherever distances are separated by more than that.
#include "tdr_common.h"

namespace tdr {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void norms_f64_kernel(const double* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                        double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    double s = 0.0;
    for (int c = lane; c < d; c += 64) { const double x = X[r * ldx + c]; s += x * x; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[r] = s;
}

struct KnnF64Params {
    const double* Xq; int64_t ldq; const double* qn; int64_t nq, q_global0;
    const double* Y; int64_t ldy; const double* yn; int64_t n_db;
    int d, k, metric, exclude_self;
    double diag_add;
    double* out_d; int32_t* out_i;    // (nq, k) top-k, or
    double* dense; int64_t ldo;       // (nq, n_db) full matrix when k == 0
};

// insertion of (v, id) into the ascending (value, index)-ordered k-list (lv, li) of one query, by the whole wavefront
__device__ __forceinline__ double list_insert_f64(double* lv, int32_t* li, int k, double v, int32_t id, int lane) {
    int pos = 0;
    for (int p0 = 0; p0 < k; p0 += 64) {
        const int p = p0 + lane;
        bool less = false;
        if (p < k) { const double ev = lv[p]; less = ev < v || (ev == v && li[p] < id); }
        pos += __popcll(__ballot(less));
    }
    if (pos >= k) return lv[k - 1];
    // shift [pos, k - 2] one slot up, highest chunk first (a chunk reads before it writes)
    for (int p0 = ((k - 1) / 64) * 64; p0 >= 0; p0 -= 64) {
        const int p = p0 + lane;
        const bool mv = p >= pos && p < k - 1;
        double ev = 0.0; int32_t ei = 0;
        if (mv) { ev = lv[p]; ei = li[p]; }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        if (mv) { lv[p + 1] = ev; li[p + 1] = ei; }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) { lv[pos] = v; li[pos] = id; }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    return lv[k - 1];
}

// One workgroup = 4 wavefronts x 16 queries; database tiles of 16 rows go through LDS and are shared by the wavefronts.
// Operands are swapped as in the float32 scan: database rows -> MFMA rows, queries -> MFMA columns, so a lane owns ONE
// query (lane & 15) and four database rows of the 16 x 16 tile: the f64 MFMA's C/D map is row = (lane >> 4) + 4 * reg
// (NOT the f32 16x16x4 map 4 * (lane >> 4) + reg).
template <int DQ>  // feature quads: d <= 4 * DQ
__global__ __launch_bounds__(256) void knn_f64_kernel(const KnnF64Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int DP = 4 * DQ + 2;  // padded row of the staged tile (conflict-free 8-byte fragment reads)
    double* tile = reinterpret_cast<double*>(smem_raw);                 // [16][DP]
    double* tyn = tile + 16 * DP;                                        // [16]
    double* lists_v = tyn + 16;                                          // [64][k]
    int32_t* lists_i = reinterpret_cast<int32_t*>(lists_v + (size_t)64 * (P.k > 0 ? P.k : 0));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const int64_t qrow = (int64_t)blockIdx.x * 64 + w * 16 + q;
    const bool qok = qrow < P.nq;
    double bq[DQ];
#pragma unroll
    for (int s = 0; s < DQ; ++s) {
        const int c = 4 * s + kq;
        bq[s] = (qok && c < P.d) ? P.Xq[qrow * P.ldq + c] : 0.0;
    }
    const double xn = qok ? P.qn[qrow] : 0.0;
    const double INF = __builtin_inf();
    if (P.k > 0) {
        for (int e = threadIdx.x; e < 64 * P.k; e += 256) { lists_v[e] = INF; lists_i[e] = 0x7fffffff; }
    }
    double tau = INF;
    const int64_t self_j = (P.exclude_self && qok) ? P.q_global0 + qrow : -1;
    const int64_t n_tiles = (P.n_db + 15) / 16;
    for (int64_t T = 0; T < n_tiles; ++T) {
        __syncthreads();
        for (int e = threadIdx.x; e < 16 * 4 * DQ; e += 256) {
            const int r = e / (4 * DQ), c = e - r * (4 * DQ);
            const int64_t row = T * 16 + r;
            tile[r * DP + c] = (row < P.n_db && c < P.d) ? P.Y[row * P.ldy + c] : 0.0;
        }
        if (threadIdx.x < 16) {
            const int64_t row = T * 16 + threadIdx.x;
            tyn[threadIdx.x] = row < P.n_db ? P.yn[row] : INF;
        }
        __syncthreads();
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < DQ; ++s) {
            const double a = tile[q * DP + 4 * s + kq];   // A[row = lane & 15][k = 4 s + (lane >> 4)]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = kq + 4 * r;
            const int64_t j = T * 16 + i;
            double c;
            if (P.metric == 2) c = -acc[r];
            else {
                c = (xn + tyn[i]) - 2.0 * acc[r];
                if (P.metric == 1) c = sqrt(fmax(c, 0.0));
            }
            const bool inside = qok && j < P.n_db;
            if (P.k == 0) {
                if (inside) {
                    if (P.exclude_self && j == P.q_global0 + qrow) c += P.diag_add;
                    P.dense[qrow * P.ldo + j] = c;
                }
                continue;
            }
            const bool hit = inside && j != self_j && c <= tau;
            unsigned long long m = __ballot(hit);
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                const int sq = src & 15;
                const double cv = __shfl(c, src, 64);
                const int32_t jv = (int32_t)(T * 16 + (src >> 4) + 4 * r);
                const double nt = list_insert_f64(lists_v + (size_t)(w * 16 + sq) * P.k, lists_i + (size_t)(w * 16 + sq) * P.k, P.k, cv,
                                                  jv, lane);
                if (q == sq) tau = nt;
            }
        }
    }
    if (P.k > 0) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * P.k; e += 256) {
            const int64_t row = (int64_t)blockIdx.x * 64 + e / P.k;
            if (row < P.nq) {
                P.out_d[row * P.k + e % P.k] = lists_v[e];
                P.out_i[row * P.k + e % P.k] = lists_i[e];
            }
        }
    }
}

// ---- root searches, one wavefront per row (rows are re-read from L2 at every evaluation) -----------------------------------
struct RowF64 { const double* c; int k, lane; };
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wmin(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double umap_f64(const RowF64& R, double rho, double target, double eps) {
    double m = -__builtin_inf();
    for (int j = R.lane; j < R.k; j += 64) m = fmax(m, (-(R.c[j] - rho)) / eps);
    m = wmax(m);
    double s = 0.0;
    for (int j = R.lane; j < R.k; j += 64) s += exp((-(R.c[j] - rho)) / eps - m);
    s = wsum(s);
    return exp(m + log(s)) - target;
}
__device__ __forceinline__ double entropic_f64(const RowF64& R, double target, double eps, double* lse_out) {
    double m = -__builtin_inf();
    for (int j = R.lane; j < R.k; j += 64) m = fmax(m, (-R.c[j]) / eps);
    m = wmax(m);
    double s = 0.0;
    for (int j = R.lane; j < R.k; j += 64) s += exp((-R.c[j]) / eps - m);
    s = wsum(s);
    const double lse = m + log(s);
    double h = 0.0;
    for (int j = R.lane; j < R.k; j += 64) { const double l = (-R.c[j]) / eps - lse; h += exp(l) * (l - 1.0); }
    h = wsum(h);
    if (lse_out) *lse_out = lse;
    return (-h) - target;
}
// root_search.py:17-77 + :147-198 (tol on |f(m)|)
template <typename F>
__device__ __forceinline__ double search_f64(F f, double b, double e, int max_iter, double tol) {
    for (int it = 0; it < max_iter; ++it) { if (!(f(b) > 0.0)) break; e = fmin(e, b); b *= 0.5; }
    for (int it = 0; it < max_iter; ++it) { if (!(f(e) < 0.0)) break; b = fmax(b, e); e *= 2.0; }
    double f_b = f(b), m = (b + e) * 0.5, f_m = f(m);
    for (int it = 0; it < max_iter; ++it) {
        if (!(fabs(f_m) >= tol)) break;
        if (f_m * f_b > 0.0) { b = m; f_b = f_m; } else e = m;
        m = (b + e) * 0.5;
        f_m = f(m);
    }
    return m;
}

__global__ __launch_bounds__(256) void umap_search_f64_kernel(const double* __restrict__ C, int64_t n, int k, double target,
                                                              int max_iter, double tol, double* __restrict__ rho_out,
                                                              double* __restrict__ eps_out, double* __restrict__ P_out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    RowF64 R{C + (size_t)row * k, k, (int)(threadIdx.x & 63)};
    double mn = __builtin_inf();
    for (int j = R.lane; j < k; j += 64) mn = fmin(mn, R.c[j]);
    const double rho = wmin(mn);
    const double eps = search_f64([&](double x) { return umap_f64(R, rho, target, x); }, 1.0, 1.0, max_iter, tol);
    for (int j = R.lane; j < k; j += 64) P_out[(size_t)row * k + j] = exp((-(R.c[j] - rho)) / eps);
    if (R.lane == 0) { rho_out[row] = rho; eps_out[row] = eps; }
}

struct EntropicScalarsF64 { double target; int use_bounds; double tN_logratio, tN_m1, log_ratio, beta_u_num, log_n; };

__global__ __launch_bounds__(256) void entropic_search_f64_kernel(const double* __restrict__ C, int64_t n, int k, EntropicScalarsF64 S,
                                                                  int max_iter, double tol, double* __restrict__ eps_out,
                                                                  double* __restrict__ lognorm_out, double* __restrict__ logP_out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    RowF64 R{C + (size_t)row * k, k, (int)(threadIdx.x & 63)};
    double b = 1.0, e = 1.0;
    if (S.use_bounds) {  // entropic.py:96-113
        double mx = -__builtin_inf(), m1 = __builtin_inf();
        for (int j = R.lane; j < k; j += 64) { mx = fmax(mx, R.c[j]); m1 = fmin(m1, R.c[j]); }
        const double dN = wmax(mx), d1 = wmin(m1);
        int first_eq = 1 << 30;
        for (int j = R.lane; j < k; j += 64) if (R.c[j] == d1) first_eq = min(first_eq, j);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) first_eq = min(first_eq, __shfl_xor(first_eq, o, 64));
        double m2 = __builtin_inf();
        for (int j = R.lane; j < k; j += 64) if (j != first_eq) m2 = fmin(m2, R.c[j]);
        const double d2 = wmin(m2);
        const double beta_L = fmax(S.tN_logratio / (S.tN_m1 * (dN - d1)), sqrt(S.log_ratio / (dN * dN - d1 * d1)));
        const double beta_U = S.beta_u_num / (d2 - d1);
        b = 1.0 / beta_U + 1e-6;
        e = 1.0 / beta_L;
    }
    const double eps = search_f64([&](double x) { return entropic_f64(R, S.target, x, nullptr); }, b, e, max_iter, tol);
    double lse;
    entropic_f64(R, S.target, eps, &lse);
    for (int j = R.lane; j < k; j += 64) logP_out[(size_t)row * k + j] = ((-R.c[j]) / eps - lse) - S.log_n;
    if (R.lane == 0) { eps_out[row] = eps; lognorm_out[row] = lse; }
}

// distance/base.py:384-398, float64; mode 0 sqeuclidean, 1 euclidean, 2 manhattan, 3 angular
__global__ __launch_bounds__(256) void indexed_sqdist_f64_kernel(const double* __restrict__ X, int64_t nx, int d, const double* __restrict__ Y,
                                                                 int64_t ny, const int64_t* __restrict__ q, int64_t nq, int nk, int mode,
                                                                 const int64_t* __restrict__ keys, double* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nq * nk) return;
    int64_t qi = q[idx / nk];
    if (qi < 0) qi += nx;
    int64_t kj = keys[idx];
    if (kj < 0) kj += ny;
    const double* x = X + (size_t)qi * d;
    const double* y = Y + (size_t)kj * d;
    double acc = 0.0;
    for (int t = 0; t < d; ++t) {
        const double df = x[t] - y[t];
        acc += mode == 2 ? fabs(df) : (mode == 3 ? x[t] * y[t] : df * df);
    }
    out[idx] = mode == 1 ? sqrt(acc) : (mode == 3 ? -acc : acc);
}

// values of the symmetrised graph on the pattern the float32 pipeline built (utils/sparse.py:138-206): entry (i, j) of
// CSR row i gets P_ij + P_ji - P_ij P_ji (mode 0) or P_ij + P_ji (mode 1), duplicates of a column summed as scatter_add does
__global__ __launch_bounds__(256) void sym_values_f64_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                             int64_t n, const int32_t* __restrict__ nn, const double* __restrict__ P,
                                                             int k, int64_t row_offset, int mode, const int64_t* __restrict__ ext_rowptr,
                                                             const int32_t* __restrict__ ext_col, const double* __restrict__ ext_val,
                                                             double* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const int64_t gi = i + row_offset;
    for (int64_t e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) {
        const int64_t j = cols[e];
        double pij = 0.0, pji = 0.0;
        for (int t = 0; t < k; ++t) if (nn[i * k + t] == j) pij += P[i * k + t];
        const int64_t lj = j - row_offset;
        if (lj >= 0 && lj < n) {
            for (int t = 0; t < k; ++t) if (nn[lj * k + t] == gi) pji += P[lj * k + t];
        } else if (ext_rowptr) {     // row j lives on another rank: its edge j -> i arrived as a transposed entry of row i
            for (int64_t t = ext_rowptr[i]; t < ext_rowptr[i + 1]; ++t) if (ext_col[t] == j) pji += ext_val[t];
        }
        vals[e] = mode == 0 ? (pij + pji) - pij * pji : pij + pji;
    }
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* LDS bytes of tdr_knn_f64 for (d, k); 0 = unsupported (d > 256, or the k-lists of 64 queries do not fit 160 KiB) */
int64_t tdr_knn_f64_lds_bytes(int d, int k) {
    if (d <= 0 || d > 256 || k < 0) return 0;
    const int dq = d <= 32 ? 8 : d <= 64 ? 16 : d <= 128 ? 32 : 64;
    const int64_t b = (int64_t)16 * (4 * dq + 2) * 8 + 16 * 8 + (int64_t)64 * k * 12;
    return b <= 160 * 1024 ? b : 0;
}

/* distance/torch.py:21-125 in float64: squared-Euclidean (0) / Euclidean (1) / angular (2) distances of the queries Xq
 * against Y.  k > 0: the k smallest per query, ascending by (distance, index) -> out_d (nq, k), out_i (nq, k) with the
 * query's own row (q_global0 + row) excluded when exclude_self; k == 0: the dense (nq, n_db) matrix into out_d (row stride
 * ldo), diag_add added to the self entries when exclude_self (:111-116).  ws: (nq + n_db) doubles (norms). */
int tdr_knn_f64(const double* Xq, int64_t nq, int64_t ldq, int64_t q_global0, const double* Y, int64_t n_db, int64_t ldy, int d, int k,
                int metric, int exclude_self, double diag_add, double* out_d, int32_t* out_i, int64_t ldo, double* ws, void* stream) {
    if (!Xq || !Y || !out_d || !ws || nq <= 0 || n_db <= 0 || d <= 0 || ldq < d || ldy < d || k < 0) return TDR_ERR_BAD_ARG;
    if (k > 0 && (!out_i || k > n_db - (exclude_self ? 1 : 0))) return TDR_ERR_BAD_ARG;
    if (k == 0 && ldo < n_db) return TDR_ERR_BAD_ARG;
    if (metric < 0 || metric > 2) return TDR_ERR_BAD_ARG;
    const int64_t lds = tdr_knn_f64_lds_bytes(d, k);
    if (lds == 0 || n_db >= 0x7fffffffLL) return TDR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    double* qn = ws;
    double* yn = ws + nq;
    hipLaunchKernelGGL(norms_f64_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, Xq, nq, d, ldq, qn);
    hipLaunchKernelGGL(norms_f64_kernel, dim3((unsigned)((n_db + 3) / 4)), dim3(256), 0, st, Y, n_db, d, ldy, yn);
    KnnF64Params P;
    P.Xq = Xq; P.ldq = ldq; P.qn = qn; P.nq = nq; P.q_global0 = q_global0; P.Y = Y; P.ldy = ldy; P.yn = yn; P.n_db = n_db;
    P.d = d; P.k = k; P.metric = metric; P.exclude_self = exclude_self; P.diag_add = diag_add;
    P.out_d = out_d; P.out_i = out_i; P.dense = out_d; P.ldo = ldo;
    const unsigned grid = (unsigned)((nq + 63) / 64);
#define TDR_KNN64(DQV)                                                                                                    \
    {                                                                                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_f64_kernel<DQV>),                            \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        if (e != hipSuccess) return (int)e;                                                                               \
        hipLaunchKernelGGL(knn_f64_kernel<DQV>, dim3(grid), dim3(256), (size_t)lds, st, P);                                \
    }
    if (d <= 32) TDR_KNN64(8)
    else if (d <= 64) TDR_KNN64(16)
    else if (d <= 128) TDR_KNN64(32)
    else TDR_KNN64(64)
#undef TDR_KNN64
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* affinity/knn_normalized.py:445-465 in float64 (see tdr_umap_search_f32) */
int tdr_umap_search_f64(const double* C, int64_t n, int k, double target, int max_iter, double tol, double* rho, double* eps,
                        double* P, void* stream) {
    if (!C || !rho || !eps || !P || n <= 0 || k <= 0) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(umap_search_f64_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, C, n, k, target, max_iter,
                       tol, rho, eps, P);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* affinity/entropic.py:272-310 in float64 (see tdr_entropic_search_f32: same scalar arguments) */
int tdr_entropic_search_f64(const double* C, int64_t n, int k, double target, double log_n, int max_iter, double tol, int use_bounds,
                            double tN, double perplexity, double p1, double* eps, double* lognorm, double* logP, void* stream) {
    if (!C || !eps || !lognorm || !logP || n <= 0 || k <= 0) return TDR_ERR_BAD_ARG;
    EntropicScalarsF64 S;
    S.target = target; S.use_bounds = use_bounds; S.log_n = log_n;
    S.log_ratio = log(tN / perplexity);
    S.tN_logratio = tN * S.log_ratio;
    S.tN_m1 = tN - 1.0;
    S.beta_u_num = log((tN - 1.0) * p1 / (1.0 - p1));
    hipLaunchKernelGGL(entropic_search_f64_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, C, n, k, S, max_iter, tol,
                       eps, lognorm, logP);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* distance/base.py:384-398 in float64; mode 0 sqeuclidean, 1 euclidean, 2 manhattan, 3 angular */
int tdr_indexed_sqdist_f64(const double* X, int64_t nx, int d, const double* Y, int64_t ny, const int64_t* q, int64_t nq, int nk, int mode,
                           const int64_t* keys, double* out, void* stream) {
    if (!X || !Y || !q || !keys || !out || nq <= 0 || nk <= 0 || d <= 0 || mode < 0 || mode > 3) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(indexed_sqdist_f64_kernel, dim3((unsigned)((nq * nk + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, nx, d, Y, ny,
                       q, nq, nk, mode, keys, out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* utils/sparse.py:138-206 in float64: values of the symmetrised graph on the CSR pattern (rowptr, cols) that
 * tdr_sym_count_f32 / tdr_sym_fill_f32 built from the same (nn, float(P)) block; mode 0 = P + P^T - P o P^T, 1 = P + P^T. */
int tdr_sym_values_f64(const int64_t* rowptr, const int32_t* cols, int64_t n, const int32_t* nn, const double* P, int k,
                       int64_t row_offset, int mode, double* vals, void* stream) {
    if (!rowptr || !cols || !nn || !P || !vals || n <= 0 || k <= 0 || mode < 0 || mode > 1) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(sym_values_f64_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rowptr, cols, n, nn, P, k,
                       row_offset, mode, (const int64_t*)nullptr, (const int32_t*)nullptr, (const double*)nullptr, vals);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* The same for one rank's rows of a row-sharded graph (utils/sparse.py:209-342): the transposed entries whose source row
 * lives on another rank come as a CSR over the LOCAL rows (ext_rowptr (n + 1), ext_col global source row, ext_val = P of
 * that edge), as parallel.exchange_transposed_edges delivers them once sorted by row. */
int tdr_sym_values_ext_f64(const int64_t* rowptr, const int32_t* cols, int64_t n, const int32_t* nn, const double* P, int k,
                           int64_t row_offset, int mode, const int64_t* ext_rowptr, const int32_t* ext_col, const double* ext_val,
                           double* vals, void* stream) {
    if (!rowptr || !cols || !nn || !P || !vals || n <= 0 || k <= 0 || mode < 0 || mode > 1) return TDR_ERR_BAD_ARG;
    if (!ext_rowptr || !ext_col || !ext_val) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(sym_values_f64_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rowptr, cols, n, nn, P, k,
                       row_offset, mode, ext_rowptr, ext_col, ext_val, vals);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
