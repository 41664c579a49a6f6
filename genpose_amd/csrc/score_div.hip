// Score and Skilling-Hutchinson divergence estimate in one launch: the right-hand side of
// cond_ode_likelihood (networks/gf_algorithms/samplers.py:41-93).  The reference evaluates the score network twice per
// ODE function call - once under no_grad for the score, once under autograd for
//     div = eps^T (d score / d x) eps = sum_j eps_j * d/dx_j sum_i score_i * eps_i            (:52-62)
// - and crosses PCIe for both.  Here one 16-row tile runs the forward trunk and the vector-Jacobian product back to back
// (score_bwd.h: score_vjp_tile).
#include "score_bwd.h"

namespace {

using namespace gp_bwd;

template <int MODE>
__global__ __launch_bounds__(DNT) void score_div_kernel(int nrows, int kcand, gp_scorenet net, const float *__restrict__ cvec,
                                                        const float *__restrict__ tvec, const float *__restrict__ x, const float *__restrict__ eps,
                                                        const float *__restrict__ sigma_dev, float *__restrict__ score, float *__restrict__ div) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int row0 = blockIdx.x * DP, tid = threadIdx.x;
    TrunkPre<DP> pre;
    trunk_begin<DP>(net, pre, cvec, tvec, row0, nrows, kcand);
    float sigma = *sigma_dev;
    gp_pin(sigma);
    load_x_tile<DP>(lds, x, row0, nrows);
    if (MODE == SCORE_DIV) load_probe_tile(lds, eps, row0, nrows);
    __syncthreads();
    const float *out = score_vjp_tile<MODE>(lds, net, cvec, tvec, row0, nrows, kcand, pre, sigma);
    for (int e = tid; e < DP * POSE; e += DNT) {
        const int r = e / POSE, j = e - r * POSE;
        if (row0 + r < nrows) score[(size_t)(row0 + r) * POSE + j] = out[r * LDS_OUT + j];
    }
    if (div && tid < DP && row0 + tid < nrows) div[row0 + tid] = out[tid * LDS_OUT + 9];
}

}  // namespace

extern "C" int gp_score_div(int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x, const float *eps,
                            const float *sigma_dev, float *score, float *div, gp_stream_t s) {
    if (nclouds < 0 || k <= 0 || !net || !cvec || !tvec || !x || !eps || !sigma_dev || !score || !div) return GP_EINVAL;
    if (!net->w_headx_t || !net->w_pose2_t || !net->w_pose0_t) return GP_EINVAL;
    const int R = nclouds * k;
    if (R == 0) return GP_OK;
    const size_t lds = LDS_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(score_div_kernel<SCORE_DIV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(score_div_kernel<ENERGY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GP_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(score_div_kernel<SCORE_DIV>, dim3((R + DP - 1) / DP), dim3(DNT), lds, (hipStream_t)s, R, k, *net, cvec, tvec, x, eps, sigma_dev, score, div);
    return gp_launch_status();
}

extern "C" int gp_energy_score(int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x, const float *sigma_dev,
                               float *score, float *energy, gp_stream_t s) {
    if (nclouds < 0 || k <= 0 || !net || !cvec || !tvec || !x || !sigma_dev || !score) return GP_EINVAL;
    if (!net->w_headx_t || !net->w_pose2_t || !net->w_pose0_t) return GP_EINVAL;
    const int R = nclouds * k;
    if (R == 0) return GP_OK;
    const size_t lds = LDS_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(score_div_kernel<ENERGY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GP_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(score_div_kernel<ENERGY>, dim3((R + DP - 1) / DP), dim3(DNT), lds, (hipStream_t)s, R, k, *net, cvec, tvec, x, x, sigma_dev, score, energy);
    return gp_launch_status();
}
