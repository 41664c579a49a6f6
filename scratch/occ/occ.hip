// occupancy probe: how many workgroups with X KB of dynamic LDS share a CU on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(int iters, float *out) {
    extern __shared__ float lds[];
    float v = threadIdx.x;
    lds[threadIdx.x] = v;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + lds[(threadIdx.x + i) & 63];
    if (v == 12345.f) out[0] = v;
}
int main() {
    float *out; hipMalloc(&out, 4);
    hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {64, 512}) {
        for (int kb : {8, 32, 40, 53, 64, 70, 72, 79, 80, 96, 128, 159}) {
            for (int wgs : {256, 512}) {
                hipLaunchKernelGGL(spin, dim3(wgs), dim3(threads), kb * 1024, 0, 20000, out);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(spin, dim3(wgs), dim3(threads), kb * 1024, 0, 20000, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("threads %d lds %3d KB wgs %d: %.3f ms\n", threads, kb, wgs, ms);
            }
        }
    }
    return 0;
}
