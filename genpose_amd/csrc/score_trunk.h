// Shared trunk of PoseScoreNet / PoseEnergyNet for one tile of pose rows (included by scorenet.hip and rk45.hip).
#pragma once
#include <stdlib.h>

#include "gp_common.h"

namespace gp_trunk {

constexpr int HID = 256, HEADS = 768, POSE = 9;

// ---------------------------------------------------------------------------------------------- trunk
// LDS layout for a P-row tile (floats):  X0 [P][16+8] | H1 [P][256+8] | H2 [P][256+8] | red [4][P][12]
template <int P>
struct TrunkLds {
    static constexpr int LD0 = 16 + GP_LD_PAD, LDH = HID + GP_LD_PAD;
    static constexpr int OFF_H1 = P * LD0, OFF_H2 = OFF_H1 + P * LDH, OFF_RED = OFF_H2 + P * LDH, TOTAL = OFF_RED + 4 * P * 12;
};

// f_theta for the tile's rows -> fout[P][9] in LDS (red area, wave 0 slot), before the output bias.
// x rows must already be in X0 (cols 0..8, zero padded to 16).  rows >= nrows are clamped duplicates.
template <int P>
__device__ __forceinline__ void trunk_ftheta(float *lds, const gp_scorenet &net, const float *__restrict__ cvec, const float *__restrict__ tvec,
                                             int row0, int nrows, int kcand) {
    using L = TrunkLds<P>;
    constexpr int PT = P / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *X0 = lds, *H1 = lds + L::OFF_H1, *H2 = lds + L::OFF_H2, *red = lds + L::OFF_RED;
    dense_to_lds<4, PT, 4, true>(X0, L::LD0, net.w_pose0, net.b_pose0, POSE, HID, H1, L::LDH);
    __syncthreads();
    dense_to_lds<4, PT, 4, true>(H1, L::LDH, net.w_pose2, net.b_pose2, HID, HID, H2, L::LDH);
    __syncthreads();
    // stacked head layer (256 -> 768) with the 256 -> 3 output layers folded into the epilogue
    int cloud[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        int r = row0 + p * 16 + (lane & 15);
        if (r >= nrows) r = nrows - 1;
        cloud[p] = r / kcand;
    }
#pragma unroll 1
    for (int h = 0; h < 3; ++h) {
        // head h owns n-chunks [16h, 16h+16); wave w takes chunks 16h + w + 4i, i = 0..3
        float part[PT][3];  // [p-chunk][component], this wave's n-chunks of head h
#pragma unroll
        for (int p = 0; p < PT; ++p) part[p][0] = part[p][1] = part[p][2] = 0.f;
        int nc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) nc[i] = 16 * h + wave + 4 * i;
        f32x4 acc[4][PT];
        mfma_tile<4, PT>(H2, L::LDH, 0, net.w_headx, HID / 16, nc, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = nc[i] * 16 + 4 * (lane >> 4);  // 0..767
            const f32x4 tv = *reinterpret_cast<const f32x4 *>(tvec + ch);
            const int chh = ch - 256 * h;
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(net.w_out + (3 * h + 0) * HID + chh);
            const f32x4 w1 = *reinterpret_cast<const f32x4 *>(net.w_out + (3 * h + 1) * HID + chh);
            const f32x4 w2 = *reinterpret_cast<const f32x4 *>(net.w_out + (3 * h + 2) * HID + chh);
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const f32x4 cv = *reinterpret_cast<const f32x4 *>(cvec + (size_t)cloud[p] * HEADS + ch);
                f32x4 v = acc[i][p] + cv + tv;
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
                part[p][0] += v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
                part[p][1] += v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
                part[p][2] += v.x * w2.x + v.y * w2.y + v.z * w2.z + v.w * w2.w;
            }
        }
        // reduce over the 4 lane groups (channels) of the wave; waves are combined below through LDS (fixed order)
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = part[p][c];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (lane < 16) red[(wave * P + p * 16 + lane) * 12 + 3 * h + c] = v;
            }
    }
    __syncthreads();
    for (int e = tid; e < P * POSE; e += 256) {
        const int r = e / POSE, j = e - r * POSE;
        const float v = ((red[(0 * P + r) * 12 + j] + red[(1 * P + r) * 12 + j]) + red[(2 * P + r) * 12 + j]) + red[(3 * P + r) * 12 + j];
        // park the result in H1 (free now): H1[r*LDH + j]
        H1[r * L::LDH + j] = v + net.b_out[j];
    }
    __syncthreads();
}

template <int P>
__device__ __forceinline__ void load_x_tile(float *lds, const float *__restrict__ x, int row0, int nrows) {
    using L = TrunkLds<P>;
    for (int e = threadIdx.x; e < P * 16; e += 256) {
        const int r = e >> 4, j = e & 15;
        int g = row0 + r;
        if (g >= nrows) g = nrows - 1;
        lds[r * L::LD0 + j] = j < POSE ? x[(size_t)g * POSE + j] : 0.f;
    }
}


template <int P>
constexpr size_t trunk_lds_bytes() {
    return (size_t)TrunkLds<P>::TOTAL * sizeof(float);
}

// Rows per workgroup tile: 32 when that still gives >= 1.5 workgroups per CU, else 16 (fills more of the 256 CUs:
// the kernels are MFMA-bound per CU, so small batches want more, smaller tiles).  GP_SCORE_P overrides (tuning).
static inline int score_tile_rows(int nrows) {
    static int forced = -1;
    if (forced < 0) {
        const char *e = getenv("GP_SCORE_P");
        forced = e ? atoi(e) : 0;
    }
    if (forced == 16 || forced == 32) return forced;
    return ((nrows + 31) / 32 >= 384) ? 32 : 16;
}

// Gram-Schmidt of pytorch3d.rotation_6d_to_matrix + GenPose's column write-back (utils/misc.py:259-265):
// b1 = a1/max(|a1|,1e-12); b2 = a2 - (b1.a2) b1; b2 /= max(|b2|,1e-12)
template <typename T>
__device__ __forceinline__ void normalize_rot6(T *v) {
    const T eps = (T)1e-12;
    T n1 = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    n1 = n1 > eps ? n1 : eps;
    T b0 = v[0] / n1, b1 = v[1] / n1, b2 = v[2] / n1;
    T d = b0 * v[3] + b1 * v[4] + b2 * v[5];
    T c0 = v[3] - d * b0, c1 = v[4] - d * b1, c2 = v[5] - d * b2;
    T n2 = sqrt(c0 * c0 + c1 * c1 + c2 * c2);
    n2 = n2 > eps ? n2 : eps;
    v[0] = b0, v[1] = b1, v[2] = b2;
    v[3] = c0 / n2, v[4] = c1 / n2, v[5] = c2 / n2;
}

}  // namespace gp_trunk
