// PoseScoreNet / PoseEnergyNet evaluation and the predictor-corrector sampler step for gfx950.
// Reference: networks/gf_algorithms/scorenet.py:178-222, energynet.py:143-198, samplers.py:102-160.
//
// Exact algebra used to cut the per-evaluation work: the first Linear of each head acts on
// cat[pts_feat(1024), t_feat(128), pose_feat(256)], so
//     W1 . total = W1p . pts_feat (once per cloud: gp_cloud_embed)
//                + W1t . t_feat   (once per time value, shared by every row: gp_time_embed)
//                + W1x . pose_feat (per row, per evaluation: here, on fp32 MFMA)
// The three heads are stacked into one 768-wide layer; their 256->3 output layers are applied in the
// accumulator epilogue (no 768-wide activation ever reaches LDS).
#include "score_trunk.h"

namespace {

using namespace gp_trunk;

// ---------------------------------------------------------------------------------------------- cloud embed
// cvec[b, 768] = Wp[768 x 1024] . pts_feat[b] + b_head : plain MFMA GEMM, 16 clouds per workgroup.
__global__ __launch_bounds__(256) void cloud_embed_kernel(int nb, gp_scorenet net, const float *__restrict__ pts_feat, float *__restrict__ cvec) {
    constexpr int K = 1024, LD = K + GP_LD_PAD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * 16;
    for (int e = tid; e < 16 * (K / 4); e += 256) {
        const int r = e / (K / 4), q = e - r * (K / 4);
        int g = r0 + r;
        if (g >= nb) g = nb - 1;
        *reinterpret_cast<f32x4 *>(lds + r * LD + 4 * q) = *reinterpret_cast<const f32x4 *>(pts_feat + (size_t)g * K + 4 * q);
    }
    __syncthreads();
    // 48 n-chunks: blockIdx.y selects a group of 16, each wave takes 4 of them
    const int ncb = blockIdx.y * 16 + wave * 4;
    int nc[4] = {ncb, ncb + 1, ncb + 2, ncb + 3};
    f32x4 acc[4][1];
    mfma_tile<4, 1>(lds, LD, 0, net.w_headp, K / 16, HEADS / 16, nc, acc);
    const int row = r0 + (lane & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = nc[i] * 16 + 4 * (lane >> 4);
        const f32x4 v = acc[i][0] + *reinterpret_cast<const f32x4 *>(net.b_head + ch);
        if (row < nb) *reinterpret_cast<f32x4 *>(cvec + (size_t)row * HEADS + ch) = v;
    }
}

// ---------------------------------------------------------------------------------------------- time embed
// tvec[i, 768] = W1t . relu(Wt1 . [sin(x), cos(x)] + bt1),  x = t * W * 2 * pi   (scorenet.py:55-64,111-116)
// w_t1 and w_headt arrive TRANSPOSED ([k][n]) so consecutive threads read consecutive words.
// grid (nt, ngroups): group g reads its nt time values at t + g * t_stride and writes tvec rows [g*nt, (g+1)*nt)
__global__ __launch_bounds__(256) void time_embed_kernel(gp_scorenet net, const float *__restrict__ t, size_t t_stride, float *__restrict__ tvec) {
    __shared__ float four[128], tf[128];
    const int tid = threadIdx.x;
    const float tv = t[(size_t)blockIdx.y * t_stride + blockIdx.x];
    tvec += (size_t)blockIdx.y * gridDim.x * HEADS;
    if (tid < 64) {
        const float xp = ((tv * net.fourier_w[tid]) * 2.0f) * 3.14159274101257324f;  // f32 evaluation order of the reference
        four[tid] = sinf(xp);
        four[tid + 64] = cosf(xp);
    }
    __syncthreads();
    if (tid < 128) {
        float acc = 0.f;
        for (int k = 0; k < 128; ++k) acc = fmaf(four[k], net.w_t1[k * 128 + tid], acc);
        tf[tid] = fmaxf(acc + net.b_t1[tid], 0.f);
    }
    __syncthreads();
    // three outputs per thread, their k-ordered fmaf chains interleaved (same arithmetic per output, 3x the ILP)
    static_assert(HEADS == 3 * 256, "one pass of three outputs per thread");
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const float *w = net.w_headt + tid;
#pragma unroll 4
    for (int k = 0; k < 128; ++k) {
        const float tk = tf[k];
        a0 = fmaf(tk, w[k * HEADS], a0);
        a1 = fmaf(tk, w[k * HEADS + 256], a1);
        a2 = fmaf(tk, w[k * HEADS + 512], a2);
    }
    float *o = tvec + (size_t)blockIdx.x * HEADS + tid;
    o[0] = a0, o[256] = a1, o[512] = a2;
}

// mode 0: score = f/(sigma+1e-7)  (scorenet.py:217);  mode 1: IP energy with s = f/sigma (energynet.py:163-185)
template <int P>
__global__ __launch_bounds__(TrunkCfg<P>::NT) void score_eval_kernel(int nrows, int kcand, gp_scorenet net, const float *__restrict__ cvec,
                                                         const float *__restrict__ tvec, const float *__restrict__ x,
                                                         const float *__restrict__ sigma_dev, int mode, float *__restrict__ out) {
    using L = TrunkLds<P>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int row0 = blockIdx.x * P, tid = threadIdx.x;
    TrunkPre<P> pre;
    trunk_begin<P>(net, pre, cvec, tvec, row0, nrows, kcand);
    float sigma = *sigma_dev;  // requested now, used after the trunk
    gp_pin(sigma);
    load_x_tile<P>(lds, x, row0, nrows);
    __syncthreads();
    trunk_ftheta<P>(lds, net, cvec, tvec, row0, nrows, kcand, pre);
    const float *F = lds + L::OFF_H1;
    if (mode == 0) {
        for (int e = tid; e < P * POSE; e += TrunkCfg<P>::NT) {
            const int r = e / POSE, j = e - r * POSE;
            if (row0 + r < nrows) out[(size_t)(row0 + r) * POSE + j] = F[r * L::LDH + j] / (sigma + 1e-7f);
        }
    } else if (tid < P) {
        const int r = tid;
        if (row0 + r < nrows) {
            const float *xr = lds + r * L::LD0;
            float er = 0.f, et = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) er += xr[j] * (F[r * L::LDH + j] / sigma);
#pragma unroll
            for (int j = 6; j < 9; ++j) et += xr[j] * (F[r * L::LDH + j] / sigma);
            out[(size_t)(row0 + r) * 2 + 0] = er;
            out[(size_t)(row0 + r) * 2 + 1] = et;
        }
    }
}

// ---------------------------------------------------------------------------------------------- PC sampler
struct PcArgs {
    int nrows, kcand, step, nsteps, nblocks;
    int bpg, rows_per_group;         // blocks / rows of one batch (group): the batch-mean gradient norm is per group
    const float *cvec, *tvec_all;    // tvec_all [nsteps][768]
    const float *sched;              // [nsteps][4]: sigma(t_i), g(t_i), step_size, sqrt(step_size)  (f32, host schedule)
    const float *z_lang, *z_pred;    // [nsteps][R][9]
    const float *centre;             // [R/k... per cloud][3]
    float *x, *mean_x, *score, *partials, *traj;  // x,mean_x,score [R,9]; partials [nsteps][nblocks]; traj [nsteps][R][9] or null
    const float *gn_ext;             // [nsteps][ngroups] or null: the batch-mean gradient norm supplied from outside (a batch that is
    int ngroups;                     //   sharded over several GPUs: the mean over ALL its rows, all-reduced between the launches)
};

// Kernel for step i (0 <= i <= nsteps):
//   i > 0      : finish step i-1 for the tile's rows (Langevin corrector + Euler-Maruyama predictor, samplers.py:129-152)
//                using score_{i-1} and the batch-mean gradient norm reduced from every block's partial sum
//   i < nsteps : evaluate score_i = s(x_i, t_i) and write this block's partial sum of |score_i|_2
//   i == nsteps: (finish only) also post-process mean_x (:157-158)
template <int P>
__global__ __launch_bounds__(TrunkCfg<P>::NT) void pc_step_kernel(PcArgs a, gp_scorenet net) {
    using L = TrunkLds<P>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float s_gn;
    const int row0 = blockIdx.x * P, tid = threadIdx.x, i = a.step;
    // GP_ABL_*: ablation builds (tuning; scratch/r2_ablation.sh, profiles/r2_sampler_ablation.txt) - what each phase of a launch costs
#ifdef GP_ABL_EMPTY  // the launch itself
    return;
#endif
#ifdef GP_ABL_NOPRO
    if (i == a.nsteps) return;
#endif
    TrunkPre<P> pre;
    GP_WG_BEGIN();
    GP_T(0);
    float sigma = 1.f;
    if (i < a.nsteps) {
        trunk_begin<P>(net, pre, a.cvec, a.tvec_all + (size_t)i * HEADS, row0, a.nrows, a.kcand);
        sigma = a.sched[(size_t)i * 4 + 0];  // requested now, used after the trunk
        gp_pin(sigma);
    }
#ifdef GP_ABL_NOPRO  // no operand loads / batch-mean reduction / PC update
    if (false) {
#else
    if (i > 0) {
#endif
        // (1) row threads request their operands first; (2) meanwhile the last wave reduces the per-block partial sums
        // of step i-1 into the batch-mean gradient norm (fixed order: deterministic); (3) one barrier, then the update.
        const bool live = row0 + tid < a.nrows;
        const int r = live ? row0 + tid : a.nrows - 1;  // rows past the end: clamped duplicates (computed, never stored)
        float xv[9], gr[9], zz1[9], zz2[9], g = 0.f, dt = 0.f, sqdt = 0.f, cen[3] = {0.f, 0.f, 0.f};
        if (tid < P) {
            const float *sc = a.sched + (size_t)(i - 1) * 4;
            g = sc[1], dt = sc[2], sqdt = sc[3];
            const float *z1 = a.z_lang + ((size_t)(i - 1) * a.nrows + r) * 9;
            const float *z2 = a.z_pred + ((size_t)(i - 1) * a.nrows + r) * 9;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                xv[j] = a.x[(size_t)r * 9 + j];
                gr[j] = a.score[(size_t)r * 9 + j];
                zz1[j] = z1[j];
                zz2[j] = z2[j];
            }
            const float *cp = a.centre + (size_t)(r / a.kcand) * 3;
            cen[0] = cp[0], cen[1] = cp[1], cen[2] = cp[2];
        }
        constexpr int LASTW = TrunkCfg<P>::NT - 64;
        if (a.gn_ext) {
            if (tid == LASTW) s_gn = a.gn_ext[(size_t)(i - 1) * a.ngroups + blockIdx.x / a.bpg];
        } else if (tid >= LASTW) {
            float s = 0.f;
            const float *pp = a.partials + (size_t)(i - 1) * a.nblocks + (size_t)(blockIdx.x / a.bpg) * a.bpg;
            for (int q = tid - LASTW; q < a.bpg; q += 64) s += pp[q];
            s = wave_sum_f32(s);
            if (tid == LASTW) s_gn = s / (float)a.rows_per_group;
        }
        __syncthreads();
        GP_T(19);
        if (tid < P) {
            const float q = 0.48f / s_gn;  // snr * sqrt(pose_dim) = 0.16 * 3
            const float lstep = 2.0f * (q * q);
            const float ns = sqrtf(2.0f * lstep);
#pragma unroll
            for (int j = 0; j < 9; ++j) xv[j] = (xv[j] + lstep * gr[j]) + ns * zz1[j];
            float n1 = sqrtf(xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2]);
            float n2 = sqrtf(xv[3] * xv[3] + xv[4] * xv[4] + xv[5] * xv[5]);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                xv[j] /= n1;
                xv[3 + j] /= n2;
            }
            const float g2 = g * g;
            float mx[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const float drift = 0.f - g2 * gr[j];  // sign as written in the reference (samplers.py:147)
                mx[j] = xv[j] + drift * dt;
                xv[j] = mx[j] + (g * sqdt) * zz2[j];
            }
            normalize_rot6(xv);
            if (live) {
                if (a.traj) {
                    float *tr = a.traj + ((size_t)(i - 1) * a.nrows + r) * 9;
#pragma unroll
                    for (int j = 0; j < 6; ++j) tr[j] = xv[j];
#pragma unroll
                    for (int j = 0; j < 3; ++j) tr[6 + j] = xv[6 + j] + cen[j];
                }
#pragma unroll
                for (int j = 0; j < 9; ++j) a.x[(size_t)r * 9 + j] = xv[j];
                if (i == a.nsteps) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) mx[6 + j] += cen[j];
                    normalize_rot6(mx);
#pragma unroll
                    for (int j = 0; j < 9; ++j) a.mean_x[(size_t)r * 9 + j] = mx[j];
                }
            }
            // hand the new state to the trunk through LDS (no global round trip)
            float *xr = lds + tid * L::LD0;
#pragma unroll
            for (int j = 0; j < 9; ++j) xr[j] = xv[j];
#pragma unroll
            for (int j = 9; j < 16; ++j) xr[j] = 0.f;
        }
        if (i == a.nsteps) return;
    } else {
        load_x_tile<P>(lds, a.x, row0, a.nrows);
    }
    __syncthreads();
    GP_T(1);
    trunk_ftheta<P>(lds, net, a.cvec, a.tvec_all + (size_t)i * HEADS, row0, a.nrows, a.kcand, pre);
    GP_T(16);
#ifdef GP_ABL_NOTAIL  // no score write / norm partial
    if (sigma != 12345.f) return;
#endif
    float *F = lds + L::OFF_H1;
    for (int e = tid; e < P * POSE; e += TrunkCfg<P>::NT) {
        const int r = e / POSE, j = e - r * POSE;
        const float v = F[r * L::LDH + j] / (sigma + 1e-7f);
        F[r * L::LDH + j] = v;
        if (row0 + r < a.nrows) a.score[(size_t)(row0 + r) * POSE + j] = v;
    }
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        for (int r = tid; r < P; r += 64) {
            if (row0 + r < a.nrows) {
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < 9; ++j) q += F[r * L::LDH + j] * F[r * L::LDH + j];
                s += sqrtf(q);
            }
        }
        s = wave_sum_f32(s);
        if (tid == 0) a.partials[(size_t)i * a.nblocks + blockIdx.x] = s;
    }
    GP_T(17);
    GP_T_FLUSH();
    GP_WG_END();
}

}  // namespace

extern "C" {

int gp_score_tile_rows(int nrows) { return score_tile_rows(nrows); }

int gp_cloud_embed(int b, const gp_scorenet *net, const float *pts_feat, float *cvec, gp_stream_t s) {
    if (b < 0 || !net || !pts_feat || !cvec) return GP_EINVAL;
    if (b == 0) return GP_OK;
    const size_t lds = (size_t)16 * (1024 + GP_LD_PAD) * sizeof(float);
    auto kern = cloud_embed_kernel;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GP_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((b + 15) / 16, 3), dim3(256), lds, (hipStream_t)s, b, *net, pts_feat, cvec);
    return gp_launch_status();
}

int gp_time_embed(int nt, const gp_scorenet *net, const float *t, float *tvec, gp_stream_t s) {
    if (nt < 0 || !net || !t || !tvec) return GP_EINVAL;
    if (nt == 0) return GP_OK;
    hipLaunchKernelGGL(time_embed_kernel, dim3(nt), dim3(256), 0, (hipStream_t)s, *net, t, (size_t)0, tvec);
    return gp_launch_status();
}

int gp_time_embed_strided(int nt, int ngroups, int64_t t_stride_floats, const gp_scorenet *net, const float *t, float *tvec, gp_stream_t s) {
    if (nt < 0 || ngroups < 0 || t_stride_floats < 0 || !net || !t || !tvec) return GP_EINVAL;
    if (nt == 0 || ngroups == 0) return GP_OK;
    hipLaunchKernelGGL(time_embed_kernel, dim3(nt, ngroups), dim3(256), 0, (hipStream_t)s, *net, t, (size_t)t_stride_floats, tvec);
    return gp_launch_status();
}

int gp_score_eval(int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x,
                  const float *sigma_dev, int mode, float *out, gp_stream_t s) {
    if (nclouds < 0 || k <= 0 || !net || !cvec || !tvec || !x || !sigma_dev || !out || (mode != 0 && mode != 1)) return GP_EINVAL;
    const int R = nclouds * k;
    if (R == 0) return GP_OK;
    const int P = score_tile_rows(R);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(score_eval_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)trunk_lds_bytes<16>()) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(score_eval_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)trunk_lds_bytes<32>()) != hipSuccess)
            return GP_ELAUNCH;
        attr_done = true;
    }
    if (P == 16)
        hipLaunchKernelGGL(score_eval_kernel<16>, dim3((R + 15) / 16), dim3(TrunkCfg<16>::NT), trunk_lds_bytes<16>(), (hipStream_t)s, R, k, *net, cvec, tvec,
                           x, sigma_dev, mode, out);
    else
        hipLaunchKernelGGL(score_eval_kernel<32>, dim3((R + 31) / 32), dim3(TrunkCfg<32>::NT), trunk_lds_bytes<32>(), (hipStream_t)s, R, k, *net, cvec, tvec,
                           x, sigma_dev, mode, out);
    return gp_launch_status();
}

int gp_pc_tile_rows(int ngroups, int nclouds_per_group, int k) {
    if (ngroups <= 0 || nclouds_per_group < 0 || k <= 0) return GP_EINVAL;
    const int rg = nclouds_per_group * k;
    int P = score_tile_rows(ngroups * rg);
    if (ngroups > 1 && rg % P != 0) P = 16;  // tiles must not straddle groups
    if (ngroups > 1 && rg % P != 0) return GP_EINVAL;
    return P;
}

int gp_pc_step_grouped(int ngroups, int nclouds_per_group, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec,
                       const float *tvec_all, const float *sched, const float *z_langevin, const float *z_predictor, const float *centre,
                       float *x, float *mean_x, float *score, float *partials, float *traj, gp_stream_t s) {
    return gp_pc_step_coupled(ngroups, nclouds_per_group, k, step, nsteps, net, cvec, tvec_all, sched, z_langevin, z_predictor, centre, x, mean_x,
                              score, partials, traj, nullptr, s);
}

int gp_pc_step_coupled(int ngroups, int nclouds_per_group, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec,
                       const float *tvec_all, const float *sched, const float *z_langevin, const float *z_predictor, const float *centre,
                       float *x, float *mean_x, float *score, float *partials, float *traj, const float *gn_ext, gp_stream_t s) {
    if (ngroups <= 0 || nclouds_per_group < 0 || k <= 0 || step < 0 || step > nsteps || !net || !cvec || !tvec_all || !sched || !z_langevin ||
        !z_predictor || !centre || !x || !mean_x || !score || !partials)
        return GP_EINVAL;
    const int rg = nclouds_per_group * k, R = ngroups * rg;
    if (R == 0) return GP_OK;
    const int P = gp_pc_tile_rows(ngroups, nclouds_per_group, k);
    if (P < 0) return P;
    PcArgs a;
    a.nrows = R, a.kcand = k, a.step = step, a.nsteps = nsteps;
    a.bpg = (rg + P - 1) / P, a.rows_per_group = rg, a.nblocks = a.bpg * ngroups;
    a.cvec = cvec, a.tvec_all = tvec_all, a.sched = sched, a.z_lang = z_langevin, a.z_pred = z_predictor, a.centre = centre;
    a.x = x, a.mean_x = mean_x, a.score = score, a.partials = partials, a.traj = traj;
    a.gn_ext = gn_ext, a.ngroups = ngroups;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(pc_step_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)trunk_lds_bytes<16>()) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(pc_step_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)trunk_lds_bytes<32>()) != hipSuccess)
            return GP_ELAUNCH;
        attr_done = true;
    }
    if (P == 16)
        hipLaunchKernelGGL(pc_step_kernel<16>, dim3(a.nblocks), dim3(TrunkCfg<16>::NT), trunk_lds_bytes<16>(), (hipStream_t)s, a, *net);
    else
        hipLaunchKernelGGL(pc_step_kernel<32>, dim3(a.nblocks), dim3(TrunkCfg<32>::NT), trunk_lds_bytes<32>(), (hipStream_t)s, a, *net);
    return gp_launch_status();
}

int gp_pc_step(int nclouds, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec, const float *tvec_all,
               const float *sched, const float *z_langevin, const float *z_predictor, const float *centre, float *x, float *mean_x,
               float *score, float *partials, float *traj, gp_stream_t s) {
    return gp_pc_step_grouped(1, nclouds, k, step, nsteps, net, cvec, tvec_all, sched, z_langevin, z_predictor, centre, x, mean_x, score,
                              partials, traj, s);
}

}  // extern "C"
