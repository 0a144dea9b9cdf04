// Shared pieces of the two-stage exact kNN (tdr_knn_screen.hip: list-keeping scan, rescoring; tdr_knn_flat.hip: threshold scan,
// list selection): the fp16-split tile image, the power-of-two scale and the worst-case error band of a screening value.
#pragma once
#include "tdr_common.h"

namespace tdr {
namespace scr {

constexpr int TILE_ROWS = 32;
constexpr uint64_t KEY_SENTINEL = 0xFF800000FFFFFFFFull;  // (+inf, 0xffffffff)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float sqrt_rn(float x) { return (float)sqrt((double)x); }

// tile image of 32 rows: ks slices x {H block, L block} of 1 KiB + 32 norms + 32 floats of padding
__host__ __device__ __forceinline__ int64_t tile16_stride_floats(int ks) { return (int64_t)ks * 512 + 64; }

// meta[0] = bits of max |x| over every element that will be packed, meta[1] = bits of max ||y||^2 (database)
// scale s = 2^(13 - floor(log2(amax))): max |s x| in [2^13, 2^14) (fp16 max 65504; l stays normal down to
// |s x| = 2^-3; below that it is subnormal or flushed, accounted for by c_den)
__device__ __forceinline__ int scale_exp(uint32_t amax_bits) {
    int ex = (int)((amax_bits >> 23) & 255u) - 127;
    if ((amax_bits & 0x7fffffffu) == 0u) ex = 13;   // all-zero data: s = 1
    if (ex < -100) ex = -100;                        // subnormal-range data: keep s finite
    return 13 - ex;
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }

// 2 * E_q (see the header): dpad = padded feature count, xn = ||x_q||^2, ymax2 = max ||y||^2, se = scale exponent.
// Terms (u = 2^-24):  3 * 2^-22            fp16-split representation of both operands + the dropped l.l' product
//                     2 (3 dpad + 16) u     fp32 accumulation of the 3*dpad products inside the matrix pipe, any
//                                           order, allowing a truncating (1 ulp) adder
//                     (dpad + 4) u          the reference's own k-ordered fp32 fma chain
//                     8 u (xn + ymax2)      norm-sum association and the final roundings
//                     2^-14 per element     l values below the fp16 normal range (covers a flush-to-zero pipe)
// One-term screening (h.h' only, `terms` = 1) replaces the first term by 2 * 2^-11 + 2^-22 and has dpad products.
__device__ __forceinline__ float screen_band(float xn, float ymax2, int dpad, int se, int terms) {
    const float u = 5.9604645e-08f;  // 2^-24
    // representation: three-term split 3 * 2^-22; one term (h.h' only) 2 * 2^-11 + 2^-22
    // two terms (h.h' + h.l': the query keeps h only, the database both halves): one operand rounded to 11 bits, the
    // other to 22: 2^-11 + 2 * 2^-22
    const float c_repr = terms == 3 ? 3.0f * 2.3841858e-07f
                         : (terms == 2 ? (4.8828125e-04f + 2.0f * 2.3841858e-07f) : (2.0f * 4.8828125e-04f + 2.3841858e-07f));
    const float nprod = (float)(terms * dpad);
    const float c_rel = 2.0f * (c_repr + 2.0f * (nprod + 16.0f) * u + (dpad + 4.0f) * u) * 1.01f;
    const float c_abs = 8.0f * u;
    const float inv_s = pow2f(-se);
    const float c_den = 2.0f * 6.1035156e-05f * sqrtf((float)dpad) * 1.01f * inv_s;
    const float nx = sqrtf(xn) * 1.0001f, ny = sqrtf(ymax2) * 1.0001f;
    const float e = c_rel * nx * ny + c_abs * (xn + ymax2) + c_den * (nx + ny);
    return 2.0f * e * 1.01f;
}

}  // namespace scr
}  // namespace tdr
