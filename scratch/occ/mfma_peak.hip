// fp32 MFMA peak probe on gfx950: v_mfma_f32_16x16x4_f32 / 32x32x2, NACC independent accumulators, W waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k16(int iters, float *out) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 1.2345f) out[0] = s;
}
template <int NACC>
__global__ void k32(int iters, float *out) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 1.2345f) out[0] = s;
}
template <class F>
static void run(const char *name, F launch, double flop_per_wave_iter, int waves_per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    launch(iters); hipDeviceSynchronize();
    hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = flop_per_wave_iter * iters * waves_per_cu * 256;
    printf("%s waves/CU %2d: %.3f ms  %.1f TFLOP/s\n", name, waves_per_cu, ms, flops / ms / 1e9);
}
int main() {
    float *out; (void)hipMalloc(&out, 4);
    for (int w : {4, 8, 16}) {
        run("16x16x4 f32, 4 acc", [&](int it) { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(64 * w), 0, 0, it, out); }, 4.0 * 4 * 2048, w);
        run("16x16x4 f32, 8 acc", [&](int it) { hipLaunchKernelGGL(k16<8>, dim3(256), dim3(64 * w), 0, 0, it, out); }, 4.0 * 8 * 2048, w);
        run("16x16x4 f32, 1 acc", [&](int it) { hipLaunchKernelGGL(k16<1>, dim3(256), dim3(64 * w), 0, 0, it, out); }, 4.0 * 1 * 2048, w);
        run("32x32x2 f32, 2 acc", [&](int it) { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(64 * w), 0, 0, it, out); }, 4.0 * 2 * 4096, w);
    }
    return 0;
}
