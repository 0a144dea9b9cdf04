// lab probe (not part of the library): semantics of v_permlane32_swap and the DPP wave reductions on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned partner32(unsigned v, int h) {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return h ? r[0] : r[1];
}
template <bool MAX>
__device__ __forceinline__ unsigned wave_red(unsigned v) {
#define STEP(CTRL, RM) { const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, RM, 0xf, false); v = MAX ? (o > v ? o : v) : (o < v ? o : v); }
    STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
#undef STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__global__ void k(const unsigned* in, unsigned* o) {
    const int lane = threadIdx.x;
    unsigned v = in[lane];
    o[lane] = partner32(v, lane >> 5);
    o[64 + lane] = wave_red<true>(v);
    o[128 + lane] = wave_red<false>(v);
}
int main() {
    unsigned h[64], out[192], *d, *o;
    int bad = 0, b1 = 0, b2 = 0, b3 = 0;
    for (int trial = 0; trial < 50; ++trial) {
        unsigned mx = 0, mn = 0xffffffffu;
        for (int i = 0; i < 64; ++i) { h[i] = (unsigned)(rand() * 2654435761u) ^ (unsigned)rand(); mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
        hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(out));
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
        hipMemcpy(out, o, sizeof(out), hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; ++i) {
            if (out[i] != h[i ^ 32]) { ++bad; ++b1; }
            if (out[64 + i] != mx) { ++bad; ++b2; }
            if (out[128 + i] != mn) { ++bad; ++b3; }
        }
        hipFree(d); hipFree(o);
    }
    printf("lane_test: %d mismatches (swap %d, max %d, min %d)\n", bad, b1, b2, b3);
    return bad != 0;
}
