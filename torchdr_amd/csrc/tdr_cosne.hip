// K10 -- COSNE: stochastic neighbour embedding in the Poincare ball, float64 like the reference.
//
// Replaces (citations under /root/reference/torchdr):
//   neighbor_embedding/cosne.py:162-193   loss = -sum_ij P_ij log Q_ij (kNN graph, gathered hyperbolic distances)
//                                                + log sum_ij Q_ij (dense N x N, diagonal included)
//                                                + lambda * mean_i (||x_i||^2 - d_H(z_i, 0)^2)^2
//                                         with Q = gamma / (d_H^2 + gamma^2) and autograd for the gradient
//   distance/torch.py:101-107, distance/base.py:392-398   d_H^2 = arccosh(1 + 2 s / ((1-|zi|^2)(1-|zj|^2)) + 1e-8)^2
//   utils/radam.py:96-167 + utils/manifold.py:207-330      Riemannian Adam on the unit ball (egrad2rgrad, expmap,
//                                                          proj, parallel transport of the first moment)
//   affinity_matcher.py:552-565           the embedding is a float64 ManifoldParameter
//
// Closed form (what autograd produces): with s = |zi - zj|^2, a = 1 - |z|^2, w = 1 + 2 s/(ai aj) + 1e-8, u = arccosh w,
//   d(u^2)/dzi = 2u / sqrt(w^2 - 1) * ( 4 (zi - zj)/(ai aj) + 4 s zi/(ai^2 aj) ).
// The N x N part never exists in memory: one thread owns a row, the columns stream through LDS in tiles, and the
// column range is split over blockIdx.y into partial sums (deterministic: no atomics) that the finishing kernel folds.
#include "tdr_common.h"

namespace tdr {

constexpr int CO_TILE = 256;
constexpr int CO_MAXC = 8;   // embedding dimensions supported (the reference's use is 2)

struct CosnePairsParams {
    const double* Z; int nc; int64_t n_total, row0, n_rows;
    double gamma;
    int n_split;
    double* part;     // (n_split, n_rows, nc + 1): [0] = sum_j Q_ij, [1..nc] = sum_{j != i} dQ_ij/dz_i
};

// value and the two gradient coefficients of d_H^2 with respect to the FIRST point:
//   grad = cdiff * (zi - zo) + czi * zi
// fp64 divisions / roots are the expensive part: one reciprocal of a_i a_o, r = sqrt(w^2 - 1) formed as
// sqrt(t (t + 2)) with t = w - 1 (no cancellation) and shared between arccosh w = log1p(t + r) and its derivative 1 / r.
template <int NC>
__device__ __forceinline__ double hyp_d2(const double (&zi)[NC], double ai, double inv_ai, const double (&zo)[NC], double ao,
                                         double& cdiff, double& czi) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) { const double df = zi[c] - zo[c]; s += df * df; }
    const double inv_den = 1.0 / (ai * ao);
    const double w = 1.0 + 2.0 * (s * inv_den) + 1e-8;
    const double t = w - 1.0;
    const double r = sqrt(t * (t + 2.0));
    const double u = log1p(t + r);
    const double du = 2.0 * u / r;
    cdiff = du * 4.0 * inv_den;
    czi = cdiff * s * inv_ai;
    return u * u;
}

template <int NC>
__global__ __launch_bounds__(CO_TILE) void cosne_pairs_kernel(const CosnePairsParams P) {
    __shared__ double Zs[CO_TILE][NC + 1];
    const int64_t r = (int64_t)blockIdx.x * CO_TILE + threadIdx.x;
    const bool live = r < P.n_rows;
    const int64_t gi = P.row0 + (live ? r : 0);
    double zi[NC], ai = 1.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) { zi[c] = P.Z[(size_t)gi * NC + c]; ai -= zi[c] * zi[c]; }
    const double inv_ai = 1.0 / ai;
    double qsum = 0.0, g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = 0.0;
    const double g2 = P.gamma * P.gamma;
    // column range of this split, in whole tiles
    const int64_t n_tiles = (P.n_total + CO_TILE - 1) / CO_TILE;
    const int64_t per = (n_tiles + P.n_split - 1) / P.n_split;
    const int64_t t0 = (int64_t)blockIdx.y * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t j = t * CO_TILE + threadIdx.x;
        __syncthreads();
        if (j < P.n_total) {
            double aj = 1.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) { const double v = P.Z[(size_t)j * NC + c]; Zs[threadIdx.x][c] = v; aj -= v * v; }
            Zs[threadIdx.x][NC] = aj;
        }
        __syncthreads();
        const int cnt = (int)((P.n_total - t * CO_TILE < CO_TILE) ? (P.n_total - t * CO_TILE) : CO_TILE);
        if (!live) continue;
        for (int jj = 0; jj < cnt; ++jj) {
            double zo[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) zo[c] = Zs[jj][c];
            double cd, cz;
            const double d2 = hyp_d2<NC>(zi, ai, inv_ai, zo, Zs[jj][NC], cd, cz);
            const double inv = 1.0 / (d2 + g2);
            qsum += P.gamma * inv;
            if (t * CO_TILE + jj != gi) {                        // the diagonal only counts in the sum
                const double dq = -P.gamma * inv * inv;
#pragma unroll
                for (int c = 0; c < NC; ++c) g[c] += dq * (cd * (zi[c] - zo[c]) + cz * zi[c]);
            }
        }
    }
    if (live) {
        double* o = P.part + ((size_t)blockIdx.y * P.n_rows + r) * (NC + 1);
        o[0] = qsum;
#pragma unroll
        for (int c = 0; c < NC; ++c) o[1 + c] = g[c];
    }
}

struct CosneFinishParams {
    const double* Z; int nc; int64_t n_total, row0, n_rows;
    const int32_t* nn; const float* Pm; int k;                       // out-edges of the chunk rows
    const int64_t* t_rowptr; const int32_t* t_src; const float* t_val;  // in-edges of the chunk rows
    const double* part; int n_split;
    const double* S;                // sum_ij Q_ij (all ranks)
    const float* x_norm;            // ||x_i||^2 of the chunk rows
    double gamma, lam, exag, rep;
    double* grad;                   // (n_rows, nc) Euclidean gradient of the chunk rows
};

template <int NC>
__global__ __launch_bounds__(256) void cosne_finish_kernel(const CosneFinishParams P) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= P.n_rows) return;
    const int64_t gi = P.row0 + r;
    double zi[NC], ai = 1.0, y = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) { zi[c] = P.Z[(size_t)gi * NC + c]; y += zi[c] * zi[c]; }
    ai -= y;
    const double inv_ai = 1.0 / ai;
    const double g2 = P.gamma * P.gamma;
    double att[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) att[c] = 0.0;
    auto edge = [&](int64_t j, double p) {
        double zo[NC], ao = 1.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { zo[c] = P.Z[(size_t)j * NC + c]; ao -= zo[c] * zo[c]; }
        double cd, cz;
        const double d2 = hyp_d2<NC>(zi, ai, inv_ai, zo, ao, cd, cz);
        const double wgt = p / (d2 + g2);                        // d(-P log Q)/d(d2)
#pragma unroll
        for (int c = 0; c < NC; ++c) att[c] += wgt * (cd * (zi[c] - zo[c]) + cz * zi[c]);
    };
    for (int e = 0; e < P.k; ++e) {                              // row end of (i -> j)
        int64_t j = P.nn[(size_t)r * P.k + e];
        if (j < 0) j += P.n_total;                               // PyTorch indexing wraps
        edge(j, (double)P.Pm[(size_t)r * P.k + e]);
    }
    for (int64_t e = P.t_rowptr[r]; e < P.t_rowptr[r + 1]; ++e)  // key end of (src -> i)
        edge(P.t_src[e], (double)P.t_val[e]);
    double rsum[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) rsum[c] = 0.0;
    for (int s = 0; s < P.n_split; ++s) {
        const double* o = P.part + ((size_t)s * P.n_rows + r) * (NC + 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) rsum[c] += o[1 + c];
    }
    // norm preservation: lam * mean_i (||x_i||^2 - h_i)^2, h = arccosh(1 + 2 y/(1-y) + 1e-8)^2
    const double w = 1.0 + 2.0 * (y / (1.0 - y)) + 1e-8;
    const double u = acosh(w);
    const double dh = (2.0 * u / sqrt(w * w - 1.0)) * (2.0 / ((1.0 - y) * (1.0 - y)));
    const double cn = P.rep * P.lam * (2.0 / (double)P.n_total) * (u * u - (double)P.x_norm[r]) * dh * 2.0;
    const double cr = P.rep * 2.0 / P.S[0];
#pragma unroll
    for (int c = 0; c < NC; ++c) P.grad[(size_t)r * NC + c] = P.exag * att[c] + cr * rsum[c] + cn * zi[c];
}

// sum over splits of part[.][r][0] -> rowsum[r]
__global__ __launch_bounds__(256) void cosne_rowsum_kernel(const double* __restrict__ part, int n_split, int64_t n_rows,
                                                          int stride, double* __restrict__ rowsum) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    double s = 0.0;
    for (int k = 0; k < n_split; ++k) s += part[((size_t)k * n_rows + r) * stride];
    rowsum[r] = s;
}

struct RadamParams {
    double* Z;            // rows to update (n_rows, nc), in place
    const double* egrad;  // Euclidean gradient of those rows
    double* exp_avg; double* exp_avg_sq;
    double* rgrad;        // optional: the Riemannian gradient (what the reference leaves in .grad)
    int64_t n_rows; int nc;
    double beta1, beta2, eps, step_size, maxnorm;
    int* nan_flag; int n_iter;
};

// One RiemannianAdam step per row on the unit ball (c = 1): utils/radam.py:139-167 with utils/manifold.py's
// egrad2rgrad / inner / expmap / proj / ptransp, in that operation order.
template <int NC>
__global__ __launch_bounds__(256) void cosne_radam_kernel(const RadamParams P) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= P.n_rows) return;
    const double MINN = 1e-15;
    double x[NC], g[NC], m[NC], v[NC], x2 = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        x[c] = P.Z[(size_t)r * NC + c]; g[c] = P.egrad[(size_t)r * NC + c];
        m[c] = P.exp_avg[(size_t)r * NC + c]; v[c] = P.exp_avg_sq[(size_t)r * NC + c];
        x2 += x[c] * x[c];
    }
    const double lam = 2.0 / fmax(1.0 - x2, MINN);
    const double lam2 = lam * lam;
    double gg = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) { g[c] = g[c] / lam2; gg += g[c] * g[c]; }            // egrad2rgrad
    const double inner = lam2 * gg;                                                      // keepdim: same for every c
    double un = 0.0, u[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        m[c] = m[c] * P.beta1 + (1.0 - P.beta1) * g[c];
        v[c] = v[c] * P.beta2 + (1.0 - P.beta2) * inner;
        u[c] = -P.step_size * (m[c] / (sqrt(v[c]) + P.eps));
        un += u[c] * u[c];
    }
    un = fmax(sqrt(un), MINN);
    // expmap: mobius_add(x, tanh(lam/2 |u|) u / |u|)
    const double th = tanh(fmin(fmax(0.5 * lam * un, -15.0), 15.0));
    double y[NC], y2 = 0.0, xy = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) { y[c] = th * u[c] / un; y2 += y[c] * y[c]; xy += x[c] * y[c]; }
    const double den = fmax(1.0 + 2.0 * xy + x2 * y2, MINN);
    double nx[NC], nn = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) { nx[c] = ((1.0 + 2.0 * xy + y2) * x[c] + (1.0 - x2) * y[c]) / den; nn += nx[c] * nx[c]; }
    nn = fmax(sqrt(nn), MINN);
    if (nn > P.maxnorm) {                                                                // proj
#pragma unroll
        for (int c = 0; c < NC; ++c) nx[c] = nx[c] / nn * P.maxnorm;
    }
    // ptransp(x, nx, m) = gyration(nx, -x, m) * lam_x / lam_nx
    double u2 = 0.0, v2 = x2, uv = 0.0, uw = 0.0, vw = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) { u2 += nx[c] * nx[c]; uv += nx[c] * (-x[c]); uw += nx[c] * m[c]; vw += (-x[c]) * m[c]; }
    const double a = -uw * v2 + vw + 2.0 * uv * vw;
    const double b = -vw * u2 - uw;
    const double d = fmax(1.0 + 2.0 * uv + u2 * v2, MINN);
    const double lam_n = 2.0 / fmax(1.0 - u2, MINN);
    bool bad = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double gy = m[c] + 2.0 * (a * nx[c] + b * (-x[c])) / d;
        P.Z[(size_t)r * NC + c] = nx[c];
        P.exp_avg[(size_t)r * NC + c] = gy * lam / lam_n;
        P.exp_avg_sq[(size_t)r * NC + c] = v[c];
        if (P.rgrad) P.rgrad[(size_t)r * NC + c] = g[c];
        bad |= !(nx[c] == nx[c]);
    }
    if (bad && P.nan_flag) atomicMax(P.nan_flag, P.n_iter + 1);
}

// one instance per embedding dimension 2 .. CO_MAXC
#define TDR_COSNE_DISPATCH(NCV, KERNEL, GRID, BLOCK, ST, ARG)                                        \
    switch (NCV) {                                                                                   \
        case 2: hipLaunchKernelGGL(KERNEL<2>, GRID, BLOCK, 0, ST, ARG); break;                       \
        case 3: hipLaunchKernelGGL(KERNEL<3>, GRID, BLOCK, 0, ST, ARG); break;                       \
        case 4: hipLaunchKernelGGL(KERNEL<4>, GRID, BLOCK, 0, ST, ARG); break;                       \
        case 5: hipLaunchKernelGGL(KERNEL<5>, GRID, BLOCK, 0, ST, ARG); break;                       \
        case 6: hipLaunchKernelGGL(KERNEL<6>, GRID, BLOCK, 0, ST, ARG); break;                       \
        case 7: hipLaunchKernelGGL(KERNEL<7>, GRID, BLOCK, 0, ST, ARG); break;                       \
        case 8: hipLaunchKernelGGL(KERNEL<8>, GRID, BLOCK, 0, ST, ARG); break;                       \
        default: return TDR_ERR_UNSUPPORTED;                                                         \
    }

}  // namespace tdr

using namespace tdr;

extern "C" {

/* column splits of the all-pairs pass (fills the chip for small N) and its scratch: n_split * n_rows * (nc + 1) doubles */
int tdr_cosne_splits(int64_t n_total, int64_t n_rows) {
    const int64_t row_blocks = (n_rows + CO_TILE - 1) / CO_TILE;
    const int64_t col_tiles = (n_total + CO_TILE - 1) / CO_TILE;
    int64_t s = (2048 + row_blocks - 1) / (row_blocks > 0 ? row_blocks : 1);
    if (s > col_tiles) s = col_tiles;
    if (s > 64) s = 64;
    return (int)(s < 1 ? 1 : s);
}

int64_t tdr_cosne_workspace_bytes(int64_t n_total, int64_t n_rows, int nc) {
    return (int64_t)tdr_cosne_splits(n_total, n_rows) * n_rows * (nc + 1) * (int64_t)sizeof(double);
}

/* Pass 1 (cosne.py:176-181): per-row sums of Q over ALL columns -> rowsum (n_rows) [the caller adds them up, across
 * ranks too, into S], and the unscaled repulsive gradient sums kept in ws for pass 2. */
int tdr_cosne_pairs_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, double gamma,
                        double* rowsum, void* ws, int64_t ws_bytes, void* stream) {
    if (!Z || !rowsum || !ws || n_total <= 0 || n_rows <= 0 || row0 < 0 || row0 + n_rows > n_total) return TDR_ERR_BAD_ARG;
    if (nc < 2 || nc > CO_MAXC) return TDR_ERR_UNSUPPORTED;
    if (ws_bytes < tdr_cosne_workspace_bytes(n_total, n_rows, nc)) return TDR_ERR_BAD_ARG;
    CosnePairsParams P;
    P.Z = Z; P.nc = nc; P.n_total = n_total; P.row0 = row0; P.n_rows = n_rows; P.gamma = gamma;
    P.n_split = tdr_cosne_splits(n_total, n_rows); P.part = (double*)ws;
    const dim3 grid((unsigned)((n_rows + CO_TILE - 1) / CO_TILE), (unsigned)P.n_split);
    hipStream_t st = (hipStream_t)stream;
    TDR_COSNE_DISPATCH(nc, cosne_pairs_kernel, grid, dim3(CO_TILE), st, P)
    TDR_CHECK_LAUNCH();
    hipLaunchKernelGGL(cosne_rowsum_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, (const double*)ws,
                       P.n_split, n_rows, nc + 1, rowsum);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Pass 2: grad (n_rows, nc) = exag * attraction (cosne.py:162-171, both ends of every kNN edge: nn / P are the chunk's
 * out-edges, t_* its in-edges as CSR) + rep * (2 / S) * repulsion sums of pass 1 + rep * lam * norm-preservation term
 * (cosne.py:183-187; x_norm = ||x_i||^2 of the chunk rows).  S: device scalar, sum of all ranks' rowsums. */
int tdr_cosne_grad_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nn,
                       const float* P_in, int k, const int64_t* t_rowptr, const int32_t* t_src, const float* t_val,
                       const double* S, const float* x_norm, double gamma, double lam, double exag, double rep,
                       const void* ws, int64_t ws_bytes, double* grad, void* stream) {
    if (!Z || !nn || !P_in || !t_rowptr || !S || !x_norm || !ws || !grad || n_rows <= 0 || k <= 0) return TDR_ERR_BAD_ARG;
    if (nc < 2 || nc > CO_MAXC) return TDR_ERR_UNSUPPORTED;
    if (ws_bytes < tdr_cosne_workspace_bytes(n_total, n_rows, nc)) return TDR_ERR_BAD_ARG;
    CosneFinishParams P;
    P.Z = Z; P.nc = nc; P.n_total = n_total; P.row0 = row0; P.n_rows = n_rows; P.nn = nn; P.Pm = P_in; P.k = k;
    P.t_rowptr = t_rowptr; P.t_src = t_src; P.t_val = t_val; P.part = (const double*)ws;
    P.n_split = tdr_cosne_splits(n_total, n_rows); P.S = S; P.x_norm = x_norm; P.gamma = gamma; P.lam = lam;
    P.exag = exag; P.rep = rep; P.grad = grad;
    const dim3 grid((unsigned)((n_rows + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    TDR_COSNE_DISPATCH(nc, cosne_finish_kernel, grid, dim3(256), st, P)
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* One RiemannianAdam step on the unit Poincare ball for n_rows rows, in place (utils/radam.py:139-167; the caller
 * supplies step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t) with the reference's doubly incremented counter t, and
 * maxnorm = 1 - 1e-5, manifold.py:233).  rgrad (optional): the rescaled gradient the reference leaves in .grad. */
int tdr_radam_poincare_f64(double* Z, const double* egrad, double* exp_avg, double* exp_avg_sq, double* rgrad,
                           int64_t n_rows, int nc, double beta1, double beta2, double eps, double step_size,
                           double maxnorm, int* nan_flag, int n_iter, void* stream) {
    if (!Z || !egrad || !exp_avg || !exp_avg_sq || n_rows <= 0) return TDR_ERR_BAD_ARG;
    if (nc < 2 || nc > CO_MAXC) return TDR_ERR_UNSUPPORTED;
    RadamParams P;
    P.Z = Z; P.egrad = egrad; P.exp_avg = exp_avg; P.exp_avg_sq = exp_avg_sq; P.rgrad = rgrad; P.n_rows = n_rows;
    P.nc = nc; P.beta1 = beta1; P.beta2 = beta2; P.eps = eps; P.step_size = step_size; P.maxnorm = maxnorm;
    P.nan_flag = nan_flag; P.n_iter = n_iter;
    const dim3 grid((unsigned)((n_rows + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    TDR_COSNE_DISPATCH(nc, cosne_radam_kernel, grid, dim3(256), st, P)
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
