// Forward + backward pass of the score trunk for one 16-row tile: the score of the network, or of the ENERGY model (the gradient of
// its inner-product energy), and the Skilling-Hutchinson divergence estimate - shared by gp_score_div / gp_energy_score
// (score_div.hip), the predictor-corrector step of the energy model (scorenet.hip) and the RK45 stage kernels that integrate the
// energy model's probability-flow ODE and the likelihood ODE (rk45.hip).
//
// The reference evaluates these with autograd (networks/gf_algorithms/energynet.py:200-222, samplers.py:49-71): a second network
// pass per evaluation.  Here one tile runs the forward trunk, keeps both hidden layers (post-ReLU) in LDS, seeds the backward pass
// on the head-layer fragments while they are in registers
//     g3[c] = [a3[c] > 0] * sum_i w_out[i][c] * u_i
// and pushes it back through the transposed weight packs (w_headx^T, w_pose2^T, w_pose0^T) on the same MFMA building block.
#pragma once
#include "score_trunk.h"

namespace gp_bwd {

using namespace gp_trunk;

constexpr int DP = 16, DNW = TrunkCfg<DP>::NW, DNV = TrunkCfg<DP>::NV, DNT = TrunkCfg<DP>::NT;
static_assert(DNW == 4 && DNV == 4, "the backward layers assume 4 waves x 4 chunks");
constexpr int LDG = HEADS + GP_LD_PAD;            // row stride of the head-layer gradient G3 [P][768]
constexpr int OFF_G3 = TrunkLds<DP, true>::TOTAL;  // after the trunk's own LDS
constexpr int OFF_U = OFF_G3 + DP * LDG;          // u [P][12] (backward seed, per mode), probe [P][12]
constexpr int LDS_FLOATS = OFF_U + 2 * DP * 12;
constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
constexpr int LDS_OUT = 12;                       // row stride of the result: out[r][0..8] = score, out[r][9] = divergence / energy

// what the tile computes from x (in X0) and the probe (in the second half of the U block)
//   SCORE_DIV : u = probe / (sigma + 1e-7);  score = f / (sigma + 1e-7);  extra = (J_f^T u) . probe   (likelihood, samplers.py:49-71)
//   ENERGY    : probe = x, u = x / sigma;    score = f / sigma + J_f^T u (= d/dx <x, f(x)/sigma>, energynet.py:200-222);
//               extra = <x, f / sigma> (the un-decoupled inner-product energy)
enum Mode { SCORE_DIV = 0, ENERGY = 1 };

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

// the probe rows of the tile -> LDS (ENERGY: the probe is x itself, taken from X0 by seed_from_x); call before the barrier that
// precedes score_vjp_tile
__device__ __forceinline__ void load_probe_tile(float *lds, const float *__restrict__ probe, int row0, int nrows) {
    float *E = lds + OFF_U + DP * 12;
    for (int e = threadIdx.x; e < DP * 12; e += DNT) {
        const int r = e / 12, j = e - r * 12;
        int g = row0 + r;
        if (g >= nrows) g = nrows - 1;
        E[e] = j < POSE ? probe[(size_t)g * POSE + j] : 0.f;
    }
}

// Preconditions: x rows in X0 (cols 0..8, zero padded to 16); MODE SCORE_DIV: probe rows loaded (load_probe_tile); ONE
// __syncthreads(); trunk_begin<DP>() issued earlier.  Result (valid after the call, which ends on a barrier): out = lds + OFF_U,
// out[r * LDS_OUT + j], j < 9 score, j == 9 divergence estimate / energy.
template <int MODE>
__device__ __forceinline__ const float *score_vjp_tile(float *lds, const gp_scorenet &net, const float *__restrict__ cvec, const float *__restrict__ tvec,
                                                       int row0, int nrows, int kcand, TrunkPre<DP> &pre, float sigma) {
    using L = TrunkLds<DP, true>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *X0 = lds, *H1 = lds + L::OFF_H1, *H2 = lds + L::OFF_H2, *G3 = lds + OFF_G3, *U = lds + OFF_U, *E = U + DP * 12;
    // backward seed u (and, ENERGY, the probe = x)
    for (int e = tid; e < DP * 12; e += DNT) {
        const int r = e / 12, j = e - r * 12;
        if (MODE == ENERGY) {
            const float xv = j < POSE ? X0[r * L::LD0 + j] : 0.f;
            E[e] = xv;
            U[e] = xv / sigma;
        } else {
            U[e] = E[e] / (sigma + 1e-7f);
        }
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
    // f_theta sits in X0 columns 12..20 (KEEP_H1).  backward: g2 = (Wx^T g3) . [h2 > 0] -> over H2;  g1 = (W2^T g2) . [h1 > 0] -> over H1
    backward_dense<HEADS / 16>(G3, LDG, net.w_headx_t, H2, L::LDH);
    __syncthreads();
    backward_dense<HID / 16>(H2, L::LDH, net.w_pose2_t, H1, L::LDH);
    __syncthreads();
    // gx = W0^T g1 (9 of 16 channels): one 16-channel chunk, wave 0; results over the (dead) seed block U
    if (wave == 0) {
        const int nc[4] = {0, 0, 0, 0};
        f32x4 acc[4][1];
        mfma_tile<1, 1>(H1, L::LDH, 0, net.w_pose0_t, HID / 16, 1, nc, acc);
        const int r = lane & 15, g = lane >> 4;
        const float *e = E + r * 12 + 4 * g;  // probe components 4g..4g+3 (zero beyond 8)
        const float gx[4] = {acc[0][0].x, acc[0][0].y, acc[0][0].z, acc[0][0].w};
        float extra = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = 4 * g + q;
            if (j < POSE) {
                if (MODE == SCORE_DIV) {
                    extra += gx[q] * e[q];
                    U[r * LDS_OUT + j] = X0[r * L::LD0 + 12 + j] / (sigma + 1e-7f);
                } else {
                    const float s = X0[r * L::LD0 + 12 + j] / sigma;
                    extra += e[q] * s;
                    U[r * LDS_OUT + j] = s + gx[q];
                }
            }
        }
        // lanes l, l+16, l+32 hold the three channel groups of row l: sum them (group 3 holds zeros)
        extra += __shfl_xor(extra, 16, 64);
        extra += __shfl_xor(extra, 32, 64);
        if (lane < 16) U[r * LDS_OUT + 9] = extra;
    }
    __syncthreads();
    return U;
}

}  // namespace gp_bwd
