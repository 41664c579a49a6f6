// Shared device helpers for libgenpose_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/genpose_hip.h"

static inline int gp_launch_status() { return hipGetLastError() == hipSuccess ? GP_OK : GP_ELAUNCH; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// fp32 MFMA building block: Y^T[N, P] = W[N, K] . X^T[K, P]  on v_mfma_f32_16x16x4_f32
// (exact f32: bitwise a k-ordered fmaf chain, 157 TFLOP/s chip peak - MI355X_MICROARCH.md).
//
//  A operand = weights.  Packed on the host (gp_pack_weight) so that ONE coalesced global_load_dwordx4
//  per lane feeds four consecutive MFMAs:
//      Wp[((nc*KG + kg)*64 + lane)*4 + jj] = W[nc*16 + (lane&15)][kg*16 + 4*(lane>>4) + jj]
//  B operand = activations in LDS, row-major [point][k] with row stride ld = Kpad + 8 floats
//  (ld = 8*odd -> the ds_read_b128 of 16 rows x 4 k-quads is bank-conflict free):
//      lane reads X[p0 + (lane&15)][kg*16 + 4*(lane>>4) .. +3]
//  MFMA #jj of k-group kg therefore contracts k = kg*16 + 4*g + jj (g = 0..3) on both operands.
//  D: lane holds Y[point p0 + (lane&15)][channel n0 + 4*(lane>>4) + r], r = 0..3  -> four consecutive
//  channels of one point: the next layer's operand is written back with one ds_write_b128.
// ------------------------------------------------------------------------------------------------
#define GP_LD_PAD 8

__host__ __device__ static inline int gp_round16(int v) { return (v + 15) & ~15; }

// One wave: NTB n-chunks x PT p-chunks of 16x16 outputs, full K loop.
//   Xs: LDS activations, ld floats per row; pc0: first p-chunk of this wave
//   Wp: packed weights; KG k-groups; nc[i]: the NTB n-chunk indices (wave-uniform; < 0 = unused)
template <int NTB, int PT>
__device__ __forceinline__ void mfma_tile(const float *Xs, int ld, int pc0, const float *__restrict__ Wp, int KG, const int (&nc)[NTB],
                                          f32x4 (&acc)[NTB][PT]) {
    const int lane = threadIdx.x & 63;
    const float *xrow[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) xrow[p] = Xs + ((pc0 + p) * 16 + (lane & 15)) * ld + 4 * (lane >> 4);
    const f32x4 *wp[NTB];
#pragma unroll
    for (int i = 0; i < NTB; ++i) wp[i] = reinterpret_cast<const f32x4 *>(Wp) + ((size_t)(nc[i] < 0 ? 0 : nc[i]) * KG) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NTB; ++i)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 wn[NTB];
#pragma unroll
    for (int i = 0; i < NTB; ++i) wn[i] = wp[i][0];
    for (int kg = 0; kg < KG; ++kg) {
        f32x4 w[NTB], x[PT];
#pragma unroll
        for (int i = 0; i < NTB; ++i) w[i] = wn[i];
        if (kg + 1 < KG) {
#pragma unroll
            for (int i = 0; i < NTB; ++i) wn[i] = wp[i][(size_t)(kg + 1) * 64];
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) x[p] = *reinterpret_cast<const f32x4 *>(xrow[p] + kg * 16);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < NTB; ++i)
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[i][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][jj], x[p][jj], acc[i][p], 0, 0, 0);
    }
}

// Dense layer over a P-point LDS tile: out = relu(X W^T + bias) written back to LDS (row stride ldo).
// Waves are arranged WN (along channels) x WP (along points); each wave owns PT p-chunks.
template <int NTB, int PT, int WN, bool RELU>
__device__ __forceinline__ void dense_to_lds(const float *Xs, int ld, const float *__restrict__ Wp, const float *__restrict__ bias, int K,
                                             int N, float *Ys, int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % WN, wp = wave / WN;
    const int KG = gp_round16(K) / 16, NC = gp_round16(N) / 16;
    const int pc0 = wp * PT;
    for (int ncb = wn; ncb < NC; ncb += WN * NTB) {
        int nc[NTB];
#pragma unroll
        for (int i = 0; i < NTB; ++i) nc[i] = (ncb + i * WN < NC) ? ncb + i * WN : -1;
        f32x4 acc[NTB][PT];
        mfma_tile<NTB, PT>(Xs, ld, pc0, Wp, KG, nc, acc);
#pragma unroll
        for (int i = 0; i < NTB; ++i) {
            if (nc[i] < 0) continue;
            const int ch = nc[i] * 16 + 4 * (lane >> 4);
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + ch);
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                f32x4 v = acc[i][p] + bv;
                if (RELU) {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f);
                    v.w = fmaxf(v.w, 0.f);
                }
                *reinterpret_cast<f32x4 *>(Ys + ((pc0 + p) * 16 + (lane & 15)) * ldo + ch) = v;
            }
        }
    }
}

// max over the 16 lanes that share (lane >> 4): the 16 points of one p-chunk
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64));
    v = fmaxf(v, __shfl_xor(v, 2, 64));
    v = fmaxf(v, __shfl_xor(v, 4, 64));
    v = fmaxf(v, __shfl_xor(v, 8, 64));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}
