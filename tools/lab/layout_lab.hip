// Lab: variants of the per-row rank sort of the loop layout (csrc/tdr_umap_sched.hip: umap_sched_layout_kernel) on a synthetic graph,
// timed with HIP events.   hipcc --offload-arch=gfx950 -O3 -o tools/layout_lab.bin tools/lab/layout_lab.hip ; ./tools/layout_lab.bin [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// V = 0: the production form (one wavefront per row, readlane rank loop); 1: no ranking (rank = lane: memory / launch floor);
// 2: the rank loop unrolled by 4 with independent accumulators
template <int V>
__global__ __launch_bounds__(256) void layout_rows(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                   const float* __restrict__ eps_per, int64_t n_rows, int32_t* __restrict__ cols_out,
                                                   float* __restrict__ eps_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int64_t b = rowptr[row], e = rowptr[row + 1];
    const int len = (int)(e - b);
    if (len > 64) return;
    const bool have = lane < len;
    const float mine = have ? eps_per[b + lane] : __builtin_inff();
    const int32_t col = have ? cols[b + lane] : 0;
    const uint32_t mbits = __float_as_uint(mine);
    const unsigned long long key = ((unsigned long long)mbits << 32) | (uint32_t)col;
    int rank = 0;
    if (V == 1) rank = lane;
    else if (V == 0) {
        for (int q = 0; q < len; ++q) {
            const unsigned long long ok = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mbits, q) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane(col, q);
            rank += (ok < key || (ok == key && q < lane)) ? 1 : 0;
        }
    } else {
        const int ulen = __builtin_amdgcn_readfirstlane(len);
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        int q = 0;
        for (; q + 4 <= ulen; q += 4) {
            unsigned long long o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                o[u] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mbits, q + u) << 32) | (uint32_t)__builtin_amdgcn_readlane(col, q + u);
            r0 += (o[0] < key || (o[0] == key && q + 0 < lane)) ? 1 : 0;
            r1 += (o[1] < key || (o[1] == key && q + 1 < lane)) ? 1 : 0;
            r2 += (o[2] < key || (o[2] == key && q + 2 < lane)) ? 1 : 0;
            r3 += (o[3] < key || (o[3] == key && q + 3 < lane)) ? 1 : 0;
        }
        for (; q < ulen; ++q) {
            const unsigned long long ok = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mbits, q) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane(col, q);
            r0 += (ok < key || (ok == key && q < lane)) ? 1 : 0;
        }
        rank = r0 + r1 + r2 + r3;
    }
    if (have) { cols_out[b + rank] = col; eps_out[b + rank] = mine; }
}

// V = 3: 16 lanes per row?  no: rows of up to 64 edges.  V = 4: a wavefront works RPW consecutive rows, the next row's loads issued
// before the current row is ranked
template <int RPW>
__global__ __launch_bounds__(256) void layout_rows_multi(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                         const float* __restrict__ eps_per, int64_t n_rows, int32_t* __restrict__ cols_out,
                                                         float* __restrict__ eps_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= n_rows) return;
    // row pointers of the wavefront's rows: lane i holds rowptr[row0 + i]
    const int64_t rp = (lane <= RPW && row0 + lane <= n_rows) ? rowptr[row0 + lane] : 0;
    int64_t b = __shfl(rp, 0, 64), e = __shfl(rp, 1, 64);
    int len = (int)(e - b);
    float mine = (lane < len && len <= 64) ? eps_per[b + lane] : __builtin_inff();
    int32_t col = (lane < len && len <= 64) ? cols[b + lane] : 0;
    for (int i = 0; i < RPW && row0 + i < n_rows; ++i) {
        // next row's loads
        int64_t nb = 0, ne = 0;
        int nlen = 0;
        float nmine = __builtin_inff();
        int32_t ncol = 0;
        if (i + 1 < RPW && row0 + i + 1 < n_rows) {
            nb = __shfl(rp, i + 1, 64); ne = __shfl(rp, i + 2, 64);
            nlen = (int)(ne - nb);
            if (lane < nlen && nlen <= 64) { nmine = eps_per[nb + lane]; ncol = cols[nb + lane]; }
        }
        if (len <= 64) {
            const uint32_t mbits = __float_as_uint(mine);
            const unsigned long long key = ((unsigned long long)mbits << 32) | (uint32_t)col;
            const int ulen = __builtin_amdgcn_readfirstlane(len);
            int rank = 0;
            for (int q = 0; q < ulen; ++q) {
                const unsigned long long ok = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mbits, q) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane(col, q);
                rank += (ok < key || (ok == key && q < lane)) ? 1 : 0;
            }
            if (lane < len) { cols_out[b + rank] = col; eps_out[b + rank] = mine; }
        }
        b = nb; len = nlen; mine = nmine; col = ncol;
    }
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    std::mt19937 gen(1);
    std::vector<int64_t> rp(n + 1, 0);
    for (int64_t i = 0; i < n; ++i) rp[i + 1] = rp[i] + 30 + (gen() % 27);
    const int64_t nnz = rp[n];
    std::vector<int32_t> cols(nnz);
    std::vector<float> eps(nnz);
    for (int64_t i = 0; i < nnz; ++i) { cols[i] = (int32_t)(gen() % n); eps[i] = 1.0f + (gen() % 100000) * 0.01f; }
    int64_t* d_rp; int32_t *d_c, *d_co; float *d_e, *d_eo;
    CK(hipMalloc(&d_rp, (n + 1) * 8)); CK(hipMalloc(&d_c, nnz * 4)); CK(hipMalloc(&d_co, nnz * 4)); CK(hipMalloc(&d_e, nnz * 4)); CK(hipMalloc(&d_eo, nnz * 4));
    CK(hipMemcpy(d_rp, rp.data(), (n + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_c, cols.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_e, eps.data(), nnz * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
        }
        // checksum of the output order
        std::vector<float> h(1 << 16); hipMemcpy(h.data(), d_eo, h.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0; for (int64_t r = 0; rp[r + 1] < (int64_t)h.size(); ++r) for (int64_t k = rp[r] + 1; k < rp[r + 1]; ++k) bad += h[k] < h[k - 1];
        printf("{\"variant\": \"%s\", \"n\": %lld, \"nnz\": %lld, \"ms\": %.3f, \"unsorted_pairs\": %d}\n", name, (long long)n, (long long)nnz, best, bad);
    };
    const dim3 g4((unsigned)((n + 3) / 4)), blk(256);
    run("production (one wavefront per row)", [&] { hipLaunchKernelGGL(layout_rows<0>, g4, blk, 0, 0, d_rp, d_c, d_e, n, d_co, d_eo); });
    run("no ranking (floor of loads, stores, launch)", [&] { hipLaunchKernelGGL(layout_rows<1>, g4, blk, 0, 0, d_rp, d_c, d_e, n, d_co, d_eo); });
    run("rank loop unrolled by 4", [&] { hipLaunchKernelGGL(layout_rows<2>, g4, blk, 0, 0, d_rp, d_c, d_e, n, d_co, d_eo); });
    run("4 rows per wavefront, next row prefetched", [&] { hipLaunchKernelGGL(layout_rows_multi<4>, dim3((unsigned)((n + 15) / 16)), blk, 0, 0, d_rp, d_c, d_e, n, d_co, d_eo); });
    run("16 rows per wavefront, next row prefetched", [&] { hipLaunchKernelGGL(layout_rows_multi<16>, dim3((unsigned)((n + 63) / 64)), blk, 0, 0, d_rp, d_c, d_e, n, d_co, d_eo); });
    return 0;
}
