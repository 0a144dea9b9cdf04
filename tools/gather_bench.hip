// Micro-benchmark: throughput of random 8-byte gathers from a region of R bytes (what bounds the UMAP negative phase).
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
    return x;
}

template <int U, int W>  // U gathers in flight per lane, W = bytes per gather (4, 8, 16)
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ Z, uint32_t n_items, int rounds, float* out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t key = mix32(tid * 0x9E3779B9u + 12345u);
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        uint32_t j[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { key = mix32(key + 0x632BE5ABu); j[u] = __umulhi(key, n_items); }
        if (W == 4) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = Z[j[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        } else if (W == 8) {
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = reinterpret_cast<const float2*>(Z)[j[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
        } else {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = reinterpret_cast<const float4*>(Z)[j[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x + v[u].w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int U, int W>
static void run(const float* Z, size_t region_bytes, int blocks, float* out) {
    const uint32_t n_items = (uint32_t)(region_bytes / W);
    const int rounds = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    gather_kernel<U, W><<<blocks, 256>>>(Z, n_items, rounds, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) gather_kernel<U, W><<<blocks, 256>>>(Z, n_items, rounds, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gathers = 5.0 * blocks * 256.0 * rounds * U;
    printf("{\"region_MiB\": %.2f, \"bytes\": %d, \"in_flight\": %d, \"blocks\": %d, \"G_gathers_per_s\": %.1f}\n",
           region_bytes / 1048576.0, W, U, blocks, gathers / (ms * 1e-3) / 1e9);
}

int main() {
    float *Z, *out;
    const size_t cap = (size_t)64 << 20;
    hipMalloc(&Z, cap); hipMemset(Z, 0, cap); hipMalloc(&out, 64);
    for (size_t mb : {1, 2, 3, 4, 8, 32}) {
        run<4, 8>(Z, mb << 20, 8192, out);
        run<8, 8>(Z, mb << 20, 8192, out);
    }
    run<4, 4>(Z, (size_t)2 << 20, 8192, out);
    run<4, 16>(Z, (size_t)4 << 20, 8192, out);
    run<2, 8>(Z, (size_t)4 << 20, 8192, out);
    run<1, 8>(Z, (size_t)4 << 20, 8192, out);
    return 0;
}
