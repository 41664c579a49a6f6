// Shared pieces of the OPT-IN split-bf16 kernels (sa_bf16x3.hip, trunk_bf16x3.hip): three bf16 matrix products with fp32 accumulation
// stand in for one fp32 product,  a . b ~= a_hi . b_hi + a_lo . b_hi + a_hi . b_lo,  x_hi = bf16(x), x_lo = bf16(x - x_hi).
#pragma once
#include "gp_common.h"

namespace gp_bf16x3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (a, b) = eight fp32 values of a lane (two D fragments: chunks 2m and 2m+1) -> the lane's eight k-values of k-block m as hi / lo vectors.
// Five VALU instructions per two values: v_cvt_pk_bf16_f32 (round to nearest even), shift / mask back to fp32, v_pk_add_f32 (x - hi is exact),
// v_cvt_pk_bf16_f32.
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8 &hi, bf16x8 &lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const bf16x2 h = __builtin_convertvector(f32x2{x[i], x[i + 1]}, bf16x2);
        const f32x2 hf = __builtin_convertvector(h, f32x2);
        const bf16x2 l = __builtin_convertvector(f32x2{x[i] - hf.x, x[i + 1] - hf.y}, bf16x2);
        hi[i] = h.x, hi[i + 1] = h.y, lo[i] = l.x, lo[i + 1] = l.y;
    }
}

__device__ __forceinline__ f32x4 relu4(const f32x4 v) { return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }

// acc += W . X with W = (wh, wl) as the A operand and X = (xh, xl) as the B operand: D lane (point, g) holds output channels 4g .. 4g+3
__device__ __forceinline__ f32x4 mma3(const bf16x8 wh, const bf16x8 wl, const bf16x8 xh, const bf16x8 xl, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc, 0, 0, 0);
}

}  // namespace gp_bf16x3
