// Score and Skilling-Hutchinson divergence estimate in one launch: the right-hand side of
// cond_ode_likelihood (networks/gf_algorithms/samplers.py:41-93).  The reference evaluates the score network twice per
// ODE function call - once under no_grad for the score, once under autograd for
//     div = eps^T (d score / d x) eps = sum_j eps_j * d/dx_j sum_i score_i * eps_i            (:52-62)
// - and crosses PCIe for both.  Here one 16-row tile runs the forward trunk, keeps both hidden layers (post-ReLU) in LDS,
// seeds the backward pass on the head-layer fragments while they are in registers
//     g3[c] = [a3[c] > 0] * sum_i w_out[i][c] * eps_i / (sigma + 1e-7)
// and pushes it back through the transposed weight packs (w_headx^T, w_pose2^T, w_pose0^T) on the same MFMA building block.
#include "score_trunk.h"

namespace {

using namespace gp_trunk;

constexpr int DP = 16, DNW = TrunkCfg<DP>::NW, DNV = TrunkCfg<DP>::NV, DNT = TrunkCfg<DP>::NT;
static_assert(DNW == 4 && DNV == 4, "the backward layers assume 4 waves x 4 chunks");
constexpr int LDG = HEADS + GP_LD_PAD;                       // row stride of the head-layer gradient G3 [P][768]
constexpr int OFF_G3 = TrunkLds<DP, true>::TOTAL;                  // after the trunk's own LDS
constexpr int OFF_U = OFF_G3 + DP * LDG;                     // u = eps / (sigma + 1e-7) [P][12], eps [P][12]
constexpr int DIV_LDS_FLOATS = OFF_U + 2 * DP * 12;

// g_out[r][n] = mask[r][n] > 0 ? sum_k Wt[n][k] * g_in[r][k] : 0 for the wave's four 16-channel chunks; written over `mask`.
template <int KG>
__device__ __forceinline__ void backward_dense(const float *Gin, int ldg, const float *__restrict__ Wt, float *MaskOut, int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc[4] = {wave, wave + DNW, wave + 2 * DNW, wave + 3 * DNW};
    f32x4 acc[4][1];
    mfma_tile<DNV, 1>(Gin, ldg, 0, Wt, KG, HID / 16, nc, acc);
#pragma unroll
    for (int i = 0; i < DNV; ++i) {
        float *m = MaskOut + (lane & 15) * ldo + nc[i] * 16 + 4 * (lane >> 4);
        const f32x4 h = *reinterpret_cast<const f32x4 *>(m);
        f32x4 g = acc[i][0];
        g.x = h.x > 0.f ? g.x : 0.f;
        g.y = h.y > 0.f ? g.y : 0.f;
        g.z = h.z > 0.f ? g.z : 0.f;
        g.w = h.w > 0.f ? g.w : 0.f;
        *reinterpret_cast<f32x4 *>(m) = g;
    }
}

// MODE 0 (likelihood):      u = eps / (sigma + 1e-7);  score = f / (sigma + 1e-7);  div = (J_f^T u) . eps
// MODE 1 (energy gradient): probe = x, u = x / sigma;   score = f / sigma + J_f^T u  (= d/dx <x, f(x)/sigma>, energynet.py:200-222);
//                           div = <x, f / sigma> (the un-decoupled IP energy)
template <int MODE>
__global__ __launch_bounds__(DNT) void score_div_kernel(int nrows, int kcand, gp_scorenet net, const float *__restrict__ cvec,
                                                        const float *__restrict__ tvec, const float *__restrict__ x, const float *__restrict__ eps,
                                                        const float *__restrict__ sigma_dev, float *__restrict__ score, float *__restrict__ div) {
    using L = TrunkLds<DP, true>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int row0 = blockIdx.x * DP, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *X0 = lds, *H1 = lds + L::OFF_H1, *H2 = lds + L::OFF_H2, *G3 = lds + OFF_G3, *U = lds + OFF_U, *E = U + DP * 12;
    TrunkPre<DP> pre;
    trunk_begin<DP>(net, pre, cvec, tvec, row0, nrows, kcand);
    float sigma = *sigma_dev;
    gp_pin(sigma);
    load_x_tile<DP>(lds, x, row0, nrows);
    for (int e = tid; e < DP * 12; e += DNT) {
        const int r = e / 12, j = e - r * 12;
        int g = row0 + r;
        if (g >= nrows) g = nrows - 1;
        const float ev = j < POSE ? eps[(size_t)g * POSE + j] : 0.f;
        E[e] = ev;
        U[e] = MODE == 0 ? ev / (sigma + 1e-7f) : ev / sigma;
    }
    __syncthreads();
    // forward; every head-layer fragment leaves its backward seed in G3 (same lane layout as the activations: one b128 store)
    trunk_ftheta<DP, true>(lds, net, cvec, tvec, row0, nrows, kcand, pre,
                           [&](int h, int, int, const f32x4 &a3, const f32x4 &w0, const f32x4 &w1, const f32x4 &w2, int ch) {
                               const float *u = U + (lane & 15) * 12 + 3 * h;
                               const float u0 = u[0], u1 = u[1], u2 = u[2];
                               f32x4 g;
                               g.x = a3.x > 0.f ? (w0.x * u0 + w1.x * u1) + w2.x * u2 : 0.f;
                               g.y = a3.y > 0.f ? (w0.y * u0 + w1.y * u1) + w2.y * u2 : 0.f;
                               g.z = a3.z > 0.f ? (w0.z * u0 + w1.z * u1) + w2.z * u2 : 0.f;
                               g.w = a3.w > 0.f ? (w0.w * u0 + w1.w * u1) + w2.w * u2 : 0.f;
                               *reinterpret_cast<f32x4 *>(G3 + (lane & 15) * LDG + ch) = g;
                           });
    // score out (f_theta parked in X0 columns 12..20 by KEEP_H1); MODE 1 adds the vector-Jacobian product at the end
    if (MODE == 0) {
        for (int e = tid; e < DP * POSE; e += DNT) {
            const int r = e / POSE, j = e - r * POSE;
            if (row0 + r < nrows) score[(size_t)(row0 + r) * POSE + j] = X0[r * L::LD0 + 12 + j] / (sigma + 1e-7f);
        }
    }
    // ---- backward: g2 = (Wx^T g3) . [h2 > 0]  -> over H2;  g1 = (W2^T g2) . [h1 > 0]  -> over H1   (trunk_ftheta ended on a barrier)
    backward_dense<HEADS / 16>(G3, LDG, net.w_headx_t, H2, L::LDH);
    __syncthreads();
    backward_dense<HID / 16>(H2, L::LDH, net.w_pose2_t, H1, L::LDH);
    __syncthreads();
    // ---- gx = W0^T g1 (9 of 16 channels), div = gx . eps : one 16-channel chunk, wave 0
    if (wave == 0) {
        const int nc[4] = {0, 0, 0, 0};
        f32x4 acc[4][1];
        mfma_tile<1, 1>(H1, L::LDH, 0, net.w_pose0_t, HID / 16, 1, nc, acc);
        const float *e = E + (lane & 15) * 12 + 4 * (lane >> 4);  // channels 4g..4g+3 (zero beyond 8)
        if (MODE == 0) {
            float d = 0.f;
            if ((lane >> 4) < 3) d = acc[0][0].x * e[0] + acc[0][0].y * e[1] + acc[0][0].z * e[2] + acc[0][0].w * e[3];
            // lanes l, l+16, l+32 hold the three channel groups of row l: sum them (group 3 holds zeros)
            d += __shfl_xor(d, 16, 64);
            d += __shfl_xor(d, 32, 64);
            if (lane < 16 && row0 + lane < nrows) div[row0 + lane] = d;
        } else {
            // score = f / sigma + J^T u per component; energy = <x, f / sigma>
            const int r = lane & 15, g = lane >> 4;
            const float gx[4] = {acc[0][0].x, acc[0][0].y, acc[0][0].z, acc[0][0].w};
            float en = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * g + q;
                if (j < POSE) {
                    const float s = X0[r * L::LD0 + 12 + j] / sigma;
                    en += e[q] * s;
                    if (row0 + r < nrows) score[(size_t)(row0 + r) * POSE + j] = s + gx[q];
                }
            }
            en += __shfl_xor(en, 16, 64);
            en += __shfl_xor(en, 32, 64);
            if (div && lane < 16 && row0 + lane < nrows) div[row0 + lane] = en;
        }
    }
}

}  // namespace

extern "C" int gp_score_div(int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x, const float *eps,
                            const float *sigma_dev, float *score, float *div, gp_stream_t s) {
    if (nclouds < 0 || k <= 0 || !net || !cvec || !tvec || !x || !eps || !sigma_dev || !score || !div) return GP_EINVAL;
    if (!net->w_headx_t || !net->w_pose2_t || !net->w_pose0_t) return GP_EINVAL;
    const int R = nclouds * k;
    if (R == 0) return GP_OK;
    const size_t lds = (size_t)DIV_LDS_FLOATS * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(score_div_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(score_div_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GP_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(score_div_kernel<0>, dim3((R + DP - 1) / DP), dim3(DNT), lds, (hipStream_t)s, R, k, *net, cvec, tvec, x, eps, sigma_dev, score, div);
    return gp_launch_status();
}

extern "C" int gp_energy_score(int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x, const float *sigma_dev,
                               float *score, float *energy, gp_stream_t s) {
    if (nclouds < 0 || k <= 0 || !net || !cvec || !tvec || !x || !sigma_dev || !score) return GP_EINVAL;
    if (!net->w_headx_t || !net->w_pose2_t || !net->w_pose0_t) return GP_EINVAL;
    const int R = nclouds * k;
    if (R == 0) return GP_OK;
    const size_t lds = (size_t)DIV_LDS_FLOATS * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(score_div_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GP_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(score_div_kernel<1>, dim3((R + DP - 1) / DP), dim3(DNT), lds, (hipStream_t)s, R, k, *net, cvec, tvec, x, x, sigma_dev, score, energy);
    return gp_launch_status();
}
