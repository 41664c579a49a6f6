// Micro-benchmark (tuning only): what does a VALU instruction beside fp32 MFMAs cost on gfx950?
//   per loop body: 16 x v_mfma_f32_16x16x4_f32 (four independent accumulators) + 16 x NV independent VALU instructions
//   KIND 0: v_fma_f32   1: v_pk_fma_f32   2: v_max_f32 (no FMA lanes)   3: v_mfma bf16 16x16x32 instead of the f32 MFMA, + v_fma_f32
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int KIND>
__global__ __launch_bounds__(256) void ub_kernel(float *out, int iters) {
    extern __shared__ float lds[];
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) ab[i] = (__bf16)a, bb[i] = (__bf16)b;
    float v[8];
    f32x2 pv[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + j, pv[j] = f32x2{v[j], v[j] + 1.f};
    const float c0 = 0.999f, c1 = 1e-3f;
    const f32x2 pc0 = {c0, c0}, pc1 = {c1, c1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if constexpr (KIND == 3)
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[m & 3], 0, 0, 0);
            else
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if constexpr (KIND == 0 || KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(c0), "v"(c1));
                if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[j & 7]) : "v"(pc0), "v"(pc1));
                if constexpr (KIND == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c1));
                if constexpr (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c0));
                if constexpr (KIND == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c1));
                if constexpr (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pv[j & 7]) : "v"(pc0));
                if constexpr (KIND == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pv[j & 7]) : "v"(pc1));
                if constexpr (KIND == 8) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c1));
                if constexpr (KIND == 9) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(c1), "v"(c0));
                if constexpr (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(v[j & 7]) : "v"(c1));
                if constexpr (KIND == 11) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(c1), "v"(c0));
                if constexpr (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j & 7]) : "v"(c1));
                if constexpr (KIND == 13) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[j & 7]));
                if constexpr (KIND == 14) asm volatile("ds_read_b128 %0, %1" : "=v"(acc[(j & 3)]) : "v"((int)threadIdx.x * 16) : "memory");
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int j = 0; j < 8; ++j) s += v[j] + pv[j].x + pv[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

template <int NV, int KIND>
static int launch(float *out, int iters, int blocks, int lds, void *st) {
    auto k = ub_kernel<NV, KIND>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, (hipStream_t)st, out, iters);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int ub_launch(int nv, int kind, float *out, int iters, int blocks, int lds, void *st) {
#define CASE(N, K) if (nv == N && kind == K) return launch<N, K>(out, iters, blocks, lds, st);
#define KINDS(N) CASE(N, 0) CASE(N, 1) CASE(N, 2) CASE(N, 3) CASE(N, 4) CASE(N, 5) CASE(N, 6) CASE(N, 7) CASE(N, 8) CASE(N, 9) CASE(N, 10) CASE(N, 11) CASE(N, 12) CASE(N, 13)
    KINDS(0) KINDS(4) KINDS(8)
    return -3;
}
