/*
 * ORACLE — TEST INFRASTRUCTURE ONLY, build container only, DIAGNOSTIC ONLY (see cuda_model.h).
 * The four `*.inc` files are the kernel bodies of the reference, extracted at test time by tests/test_cu_text_emulation.py
 * (EMU_INC_DIR is a temporary directory; nothing from /root/reference is committed).  The launch dispatch below restates the
 * launchers (which are `<<< >>>` syntax and cannot be compiled by a host compiler): sampling_gpu.cu:211-253 (block size =
 * opt_n_threads(n), one block per cloud), ball_query_gpu.cu:48-67 and interpolate_gpu.cu:55-74,99-117 (THREADS_PER_BLOCK
 * threads, DIVUP(m, THREADS_PER_BLOCK) x b blocks / x c x b).
 */
#include "cuda_model.h"
#include EMU_CUDA_UTILS_H /* the reference's own cuda_utils.h (opt_n_threads, THREADS_PER_BLOCK, DIVUP), included where it lies */

#include "fps_kernel.inc"
#include "ball_query_kernel.inc"
#include "three_nn_kernel.inc"
#include "three_interpolate_kernel.inc"

template <unsigned BS>
static void fps_block(int b, int n, int m, const float *dataset, float *temp, int *idxs) {
    for (int bi = 0; bi < b; ++bi)
        emu_run_block(bi, 0, 0, BS, [&] { furthest_point_sampling_kernel<BS>(b, n, m, dataset, temp, idxs); });
}

extern "C" {

int emu_opt_n_threads(int n) { return opt_n_threads(n); }

void emu_fps(int b, int n, int m, const float *dataset, float *temp, int *idxs) {
    gridDim = {(unsigned)b, 1, 1};
    switch (opt_n_threads(n)) {
        case 1024: fps_block<1024>(b, n, m, dataset, temp, idxs); break;
        case 512: fps_block<512>(b, n, m, dataset, temp, idxs); break;
        case 256: fps_block<256>(b, n, m, dataset, temp, idxs); break;
        case 128: fps_block<128>(b, n, m, dataset, temp, idxs); break;
        case 64: fps_block<64>(b, n, m, dataset, temp, idxs); break;
        case 32: fps_block<32>(b, n, m, dataset, temp, idxs); break;
        case 16: fps_block<16>(b, n, m, dataset, temp, idxs); break;
        case 8: fps_block<8>(b, n, m, dataset, temp, idxs); break;
        case 4: fps_block<4>(b, n, m, dataset, temp, idxs); break;
        case 2: fps_block<2>(b, n, m, dataset, temp, idxs); break;
        case 1: fps_block<1>(b, n, m, dataset, temp, idxs); break;
        default: fps_block<512>(b, n, m, dataset, temp, idxs);
    }
}

void emu_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx) {
    gridDim = {(unsigned)DIVUP(m, THREADS_PER_BLOCK), (unsigned)b, 1};
    for (unsigned by = 0; by < gridDim.y; ++by)
        for (unsigned bx = 0; bx < gridDim.x; ++bx)
            emu_run_block(bx, by, 0, THREADS_PER_BLOCK, [&] { ball_query_kernel_fast(b, n, m, radius, nsample, new_xyz, xyz, idx); });
}

void emu_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx) {
    gridDim = {(unsigned)DIVUP(n, THREADS_PER_BLOCK), (unsigned)b, 1};
    for (unsigned by = 0; by < gridDim.y; ++by)
        for (unsigned bx = 0; bx < gridDim.x; ++bx)
            emu_run_block(bx, by, 0, THREADS_PER_BLOCK, [&] { three_nn_kernel_fast(b, n, m, unknown, known, dist2, idx); });
}

void emu_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx, const float *weight, float *out) {
    gridDim = {(unsigned)DIVUP(n, THREADS_PER_BLOCK), (unsigned)c, (unsigned)b};
    for (unsigned bz = 0; bz < gridDim.z; ++bz)
        for (unsigned by = 0; by < gridDim.y; ++by)
            for (unsigned bx = 0; bx < gridDim.x; ++bx)
                emu_run_block(bx, by, bz, THREADS_PER_BLOCK, [&] { three_interpolate_kernel_fast(b, c, m, n, points, idx, weight, out); });
}

}  /* extern "C" */
