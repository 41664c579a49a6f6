// fp32 MFMA throughput with operands that actually toggle (random values, rotated every instruction) vs constant operands:
// the chip's power management lowers the clock under real data (MI355X_MICROARCH.md: zero-filled inputs ran +19 %).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool RANDOM>
__global__ void k16(int iters, const float *src, float *out) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = RANDOM ? src[(threadIdx.x * 16 + i) & 4095] : 0.f;
        b[i] = RANDOM ? src[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 4095] : 0.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + r) & 7], b[(i + 2 * r + 1) & 7], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 1.2345f) out[0] = s;
}
int main() {
    float *out, *src; (void)hipMalloc(&out, 4); (void)hipMalloc(&src, 4096 * 4);
    float h[4096]; unsigned x = 12345; for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rnd = 0; rnd < 2; ++rnd) for (int w : {4, 8}) {
        const int iters = 40000;
        auto launch = [&]() { if (rnd) hipLaunchKernelGGL(k16<true>, dim3(256), dim3(64 * w), 0, 0, iters, src, out); else hipLaunchKernelGGL(k16<false>, dim3(256), dim3(64 * w), 0, 0, iters, src, out); };
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s operands, %d waves/CU: %.3f ms  %.1f TFLOP/s\n", rnd ? "random  " : "constant", w, ms, 4.0 * 8 * 2048 * iters * w * 256 / ms / 1e9);
    }
    return 0;
}
