// Common device helpers for the torchdr_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TDR_OK 0
#define TDR_ERR_BAD_ARG (-1)
#define TDR_ERR_UNSUPPORTED (-2)
#define TDR_ERR_WORKSPACE (-3)

#define TDR_WAVE 64

#define TDR_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

namespace tdr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Monotone float -> uint32 map (total order incl. negatives), and its inverse.
__device__ __forceinline__ uint32_t f2u(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float u2f(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}
__device__ __forceinline__ uint64_t mkkey(float d, uint32_t idx) {
    return ((uint64_t)f2u(d) << 32) | (uint64_t)idx;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Wave-wide reductions through DPP-lowered shuffles (64-wide wavefront).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The same reductions on the DPP path (no LDS traffic): quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror
// cover groups of up to 16 lanes (groups are aligned to their size); wider groups finish with shuffles.
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int G>
__device__ __forceinline__ float group_sum_dpp(float v) {
    if (G >= 2) v += dpp_mov_f<0xB1>(v);
    if (G >= 4) v += dpp_mov_f<0x4E>(v);
    if (G >= 8) v += dpp_mov_f<0x141>(v);
    if (G >= 16) v += dpp_mov_f<0x140>(v);
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
template <int G>
__device__ __forceinline__ int group_sum_dpp(int v) {
    if (G >= 2) v += dpp_mov_i<0xB1>(v);
    if (G >= 4) v += dpp_mov_i<0x4E>(v);
    if (G >= 8) v += dpp_mov_i<0x141>(v);
    if (G >= 16) v += dpp_mov_i<0x140>(v);
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int G>
__device__ __forceinline__ float group_min(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace tdr
