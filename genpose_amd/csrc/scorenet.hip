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
#include "score_bwd.h"
#include "trunk_chain_vjp.h"

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

// chain form (trunk_chain.h): one wave per 16 rows, the row's pose and outputs stay in its lanes
template <int PT>
__global__ __launch_bounds__(gp_chain::NT, 1) void score_eval_chain_kernel(int nrows, int kcand, gp_scorenet net, const float *__restrict__ cvec,
                                                                                                 const float *__restrict__ tvec, const float *__restrict__ x,
                                                                                                 const float *__restrict__ sigma_dev, int mode, float *__restrict__ out) {
    using C = gp_chain::Cfg<PT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pt = lane & 15, g = lane >> 4;
    const int wg_row0 = blockIdx.x * C::ROWS;
    gp_chain::State<PT> st;
    gp_chain::begin<PT>(st, lds, net, cvec, tvec, wg_row0, nrows, kcand);
    const float sigma = *sigma_dev;
    float xv[PT][POSE];
    f32x4 xf[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        int r = wg_row0 + (wave * PT + p) * 16 + pt;
        r = r < nrows ? r : nrows - 1;  // rows past the end: clamped duplicates (computed, never stored)
#pragma unroll
        for (int j = 0; j < POSE; ++j) xv[p][j] = x[(size_t)r * POSE + j];
        xf[p] = gp_chain::pose_fragment(xv[p], g);
    }
    float f[PT][POSE];
    gp_chain::run<PT>(st, lds, net, xf, f);
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int r = wg_row0 + (wave * PT + p) * 16 + pt;
        if (r >= nrows || g != 0) continue;
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < POSE; ++j) out[(size_t)r * POSE + j] = f[p][j] / (sigma + 1e-7f);
        } else {
            float er = 0.f, et = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) er += xv[p][j] * (f[p][j] / sigma);
#pragma unroll
            for (int j = 6; j < 9; ++j) et += xv[p][j] * (f[p][j] / sigma);
            out[(size_t)r * 2 + 0] = er;
            out[(size_t)r * 2 + 1] = et;
        }
    }
}

// ---------------------------------------------------------------------------------------------- PC sampler
struct PcArgs {
    int nrows, kcand, step, nsteps;
    int nparts, ppg, rows_per_group;  // partial sums of |score| per step / per batch (group): the batch-mean gradient norm is per group
    int wgpg;                         // workgroups per group
    const float *cvec, *tvec_all;    // tvec_all [nsteps][768]
    const float *sched;              // [nsteps][4]: sigma(t_i), g(t_i), step_size, sqrt(step_size)  (f32, host schedule)
    const float *z_lang, *z_pred;    // [nsteps][R][9]
    const float *centre;             // [R/k... per cloud][3]
    float *x, *mean_x, *score, *partials, *traj;  // x,mean_x,score [R,9]; partials [nsteps][nparts]; traj [nsteps][R][9] or null
    const float *gn_ext;             // [nsteps][ngroups] or null: the batch's gradient-norm statistic supplied from outside (a batch that is
    int ngroups;                     //   sharded over several GPUs, all-reduced between the launches): the SUM of |score| over all its
    float gn_rows;                   //   rows when gn_rows > 0 (= that row count), else the mean itself
    // head-split plan (GP_PLAN_HEADSPLIT): workgroup 3 t + h evaluates head h of 16-row tile t and owns components 3 h .. 3 h + 2 of the
    // score.  THREE workgroups read a tile's state and score and each writes a part of them, so nothing a launch reads may be written by
    // the same launch: every step keeps its own copies in its row of `partials` (nparts = 21 * nrows floats per step):
    //     [0, 3R)     sum of squares of row r's three components of head h at 3 r + h (a row's norm needs all nine: the NEXT launch puts
    //                 sqrt(p[3r] + p[3r+1] + p[3r+2]) together and reduces it over its batch's rows)
    //     [3R, 12R)   score_i [R][9], read by launch i + 1
    //     [12R, 21R)  the state after launch i's update [R][9], read by launch i + 1 (launch 1 reads the initial state from `x`, which
    //                 this plan never writes)
    // (wgpg counts TILES per group.)
};

// Kernel for step i (0 <= i <= nsteps):
//   i > 0      : finish step i-1 for the tile's rows (Langevin corrector + Euler-Maruyama predictor, samplers.py:129-152)
//                using score_{i-1} and the batch-mean gradient norm reduced from every block's partial sum
//   i < nsteps : evaluate score_i = s(x_i, t_i) and write this block's partial sum of |score_i|_2
//   i == nsteps: (finish only) also post-process mean_x (:157-158)
// MODEL 0: the score network (score = f / (sigma + 1e-7), scorenet.py:217).  MODEL 1: the ENERGY network, whose score is the
// gradient of its inner-product energy (energynet.py:200-222): forward + vector-Jacobian product in the tile (score_bwd.h), 16-row
// tiles only.  The same kernel otherwise: sampling from the energy model is the same captured launch chain.
// SPLIT: the head-split plan (see PcArgs): three workgroups per tile.  Each finishes step i-1 for all 16 rows (the update is row-local
// and cheap; identical in the three, workgroup h = 0 stores) and evaluates ONE head of the score at t_i.
template <int P, int MODEL, bool SPLIT = false>
__global__ __launch_bounds__(TrunkCfg<P>::NT) void pc_step_kernel(PcArgs a, gp_scorenet net) {
    static_assert(MODEL == 0 || P == gp_bwd::DP, "the backward pass runs on 16-row tiles");
    static_assert(!SPLIT || (MODEL == 0 && P == 16), "head-split: score model, 16-row tiles");
    using L = TrunkLds<P, MODEL == 1>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float s_gn;
    const int tile = SPLIT ? blockIdx.x / 3 : blockIdx.x, hsel = SPLIT ? blockIdx.x - 3 * tile : 0;
    const int row0 = tile * P, tid = threadIdx.x, i = a.step;
    // SPLIT: three workgroups READ a tile's state and score and each writes a part - in place that would be a race between workgroups
    // (a workgroup dispatched late would read what a sibling has already written).  Launch i reads step i-1's copies and writes its own
    // (PcArgs).  The finish-only launch needs one workgroup per tile.
    if (SPLIT && i == a.nsteps && hsel != 0) return;
    const size_t R = (size_t)a.nrows;
    const float *prev = SPLIT && i > 0 ? a.partials + (size_t)(i - 1) * a.nparts : nullptr;
    float *mine = SPLIT && i < a.nsteps ? a.partials + (size_t)i * a.nparts : nullptr;
    const float *x_in = SPLIT ? (i <= 1 ? a.x : prev + 12 * R) : a.x;
    const float *score_in = SPLIT && i > 0 ? prev + 3 * R : a.score;
    float *x_out = SPLIT ? (mine ? mine + 12 * R : nullptr) : a.x;
    TrunkPre<P> pre;
    float sigma = 1.f;
    if (i < a.nsteps) {
        trunk_begin<P>(net, pre, a.cvec, a.tvec_all + (size_t)i * HEADS, row0, a.nrows, a.kcand);
        sigma = a.sched[(size_t)i * 4 + 0];  // requested now, used after the trunk
        gp_pin(sigma);
    }
    if (i > 0) {
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
                xv[j] = x_in[(size_t)r * 9 + j];
                gr[j] = score_in[(size_t)r * 9 + j];
                zz1[j] = z1[j];
                zz2[j] = z2[j];
            }
            const float *cp = a.centre + (size_t)(r / a.kcand) * 3;
            cen[0] = cp[0], cen[1] = cp[1], cen[2] = cp[2];
        }
        constexpr int LASTW = TrunkCfg<P>::NT - 64;
        if (a.gn_ext) {
            if (tid == LASTW) {
                const float v = a.gn_ext[(size_t)(i - 1) * a.ngroups + tile / a.wgpg];
                s_gn = a.gn_rows > 0.f ? v / a.gn_rows : v;
            }
        } else if (tid >= LASTW) {
            float s = 0.f;
            const float *pp = a.partials + (size_t)(i - 1) * a.nparts + (size_t)(tile / a.wgpg) * (SPLIT ? 3 * a.rows_per_group : a.ppg);
            if constexpr (SPLIT) {
                for (int r = tid - LASTW; r < a.rows_per_group; r += 64) s += sqrtf((pp[3 * r] + pp[3 * r + 1]) + pp[3 * r + 2]);
            } else {
                for (int q = tid - LASTW; q < a.ppg; q += 64) s += pp[q];
            }
            s = wave_sum_f32(s);
            if (tid == LASTW) s_gn = s / (float)a.rows_per_group;
        }
        __syncthreads();
        if (tid < P) {
            float mx[9];
            pc_update_row(xv, gr, zz1, zz2, s_gn, g, dt, sqdt, mx);
            if (live && hsel == 0) {
                if (a.traj) {
                    float *tr = a.traj + ((size_t)(i - 1) * a.nrows + r) * 9;
#pragma unroll
                    for (int j = 0; j < 6; ++j) tr[j] = xv[j];
#pragma unroll
                    for (int j = 0; j < 3; ++j) tr[6 + j] = xv[6 + j] + cen[j];
                }
                if (x_out) {
#pragma unroll
                    for (int j = 0; j < 9; ++j) x_out[(size_t)r * 9 + j] = xv[j];
                }
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
    float *F;
    int ldf;
    if constexpr (MODEL == 0) {
        trunk_ftheta<P, false, TrunkNoEmit, SPLIT>(lds, net, a.cvec, a.tvec_all + (size_t)i * HEADS, row0, a.nrows, a.kcand, pre, TrunkNoEmit(), hsel);
        F = lds + L::OFF_H1, ldf = L::LDH;
        for (int e = tid; e < P * POSE; e += TrunkCfg<P>::NT) {
            const int r = e / POSE, j = e - r * POSE;
            if (SPLIT && j / 3 != hsel) continue;
            const float v = F[r * ldf + j] / (sigma + 1e-7f);
            F[r * ldf + j] = v;
            if (row0 + r < a.nrows) {
                a.score[(size_t)(row0 + r) * POSE + j] = v;  // (an output only under the head-split plan: nothing reads it back)
                if (SPLIT) mine[3 * R + (size_t)(row0 + r) * POSE + j] = v;
            }
        }
    } else {
        F = const_cast<float *>(gp_bwd::score_vjp_tile<gp_bwd::ENERGY>(lds, net, a.cvec, a.tvec_all + (size_t)i * HEADS, row0, a.nrows, a.kcand, pre, sigma));
        ldf = gp_bwd::LDS_OUT;
        for (int e = tid; e < P * POSE; e += TrunkCfg<P>::NT) {
            const int r = e / POSE, j = e - r * POSE;
            if (row0 + r < a.nrows) a.score[(size_t)(row0 + r) * POSE + j] = F[r * ldf + j];
        }
    }
    __syncthreads();
    if constexpr (SPLIT) {
        // per row: the sum of squares of this head's three components (the row's norm is put together by the next launch)
        if (tid < P && row0 + tid < a.nrows) {
            const float *f = F + tid * ldf + 3 * hsel;
            mine[3 * (size_t)(row0 + tid) + hsel] = (f[0] * f[0] + f[1] * f[1]) + f[2] * f[2];
        }
        return;
    }
    if (tid < 64) {
        float s = 0.f;
        for (int r = tid; r < P; r += 64) {
            if (row0 + r < a.nrows) {
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < 9; ++j) q += F[r * ldf + j] * F[r * ldf + j];
                s += sqrtf(q);
            }
        }
        s = wave_sum_f32(s);
        if (tid == 0) a.partials[(size_t)i * a.nparts + blockIdx.x] = s;
    }
}

// The same launch in the chain form (trunk_chain.h).  A wave owns 16 * PT rows from the sampler update to the score: every lane
// of a row's four lane groups carries the row's 9-vector (the update is ~150 VALU instructions per wave, computed redundantly by
// the four groups - cheaper than any exchange), lane group 0 stores.  One partial sum of |score| per WAVE; the batch-mean
// gradient norm is reduced from them by every wave in the same fixed order (identical in all waves: deterministic).
// MODEL 1: the ENERGY network - its score, the gradient of the inner-product energy, from the forward + vector-Jacobian chain of
// trunk_chain_vjp.h (cotangent u = x / sigma; score = f / sigma + J_f^T u, energynet.py:200-222) - what pc_step_kernel<16, 1> computes
// per 16-row tile through LDS.
template <int PT, int MODEL>
__global__ __launch_bounds__(gp_chain::NT, 1) void pc_step_chain_kernel(PcArgs a, gp_scorenet net) {
    using C = gp_chain::Cfg<PT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), pt = lane & 15, g = lane >> 4, i = a.step;
    const int wg_row0 = blockIdx.x * C::ROWS;
    gp_chain::State<PT> st;
    const float *tvec = a.tvec_all + (size_t)(i < a.nsteps ? i : 0) * HEADS;
    int row[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) row[p] = wg_row0 + (wave * PT + p) * 16 + pt;
    // ---- the rows' operands are requested first, the ring prologue behind them: one memory round trip covers both
    float xv[PT][9], gr[PT][9], zz1[PT][9], zz2[PT][9], cen[PT][3];
    float gdiff = 0.f, dt = 0.f, sqdt = 0.f, gn = 1.f, sigma = 1.f;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int r = row[p] < a.nrows ? row[p] : a.nrows - 1;  // rows past the end: clamped duplicates (computed, never stored)
#pragma unroll
        for (int j = 0; j < 9; ++j) xv[p][j] = a.x[(size_t)r * 9 + j];
        if (i > 0) {
            const float *z1 = a.z_lang + ((size_t)(i - 1) * a.nrows + r) * 9;
            const float *z2 = a.z_pred + ((size_t)(i - 1) * a.nrows + r) * 9;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                gr[p][j] = a.score[(size_t)r * 9 + j];
                zz1[p][j] = z1[j];
                zz2[p][j] = z2[j];
            }
            const float *cp = a.centre + (size_t)(r / a.kcand) * 3;
            cen[p][0] = cp[0], cen[p][1] = cp[1], cen[p][2] = cp[2];
        }
    }
    // memory returns in order: what the sampler update needs (row operands above, schedule, the batch's partial sums) is asked for
    // first, the ring start-up behind it - the update then runs while the weights are still on their way
    float psum[4] = {0.f, 0.f, 0.f, 0.f};
    const int grp = blockIdx.x / a.wgpg;
    const float *pp = a.partials + (size_t)(i > 0 ? i - 1 : 0) * a.nparts + (size_t)grp * a.ppg;
    if (i > 0) {
        const float *sc = a.sched + (size_t)(i - 1) * 4;
        gdiff = sc[1], dt = sc[2], sqdt = sc[3];
        if (a.gn_ext) {
            gn = a.gn_ext[(size_t)(i - 1) * a.ngroups + grp];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) psum[u] = lane + 64 * u < a.ppg ? pp[lane + 64 * u] : 0.f;
        }
    }
    gp_chain::Staged<PT> sg;
    if (i < a.nsteps) {
        sigma = a.sched[(size_t)i * 4 + 0];
        gp_chain::begin_request<PT>(st, sg, net, a.cvec, tvec, wg_row0, a.nrows, a.kcand);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (i > 0) {
        if (a.gn_ext) {
            if (a.gn_rows > 0.f) gn = gn / a.gn_rows;
        } else {
            float s = ((psum[0] + psum[1]) + psum[2]) + psum[3];  // the order of `for (q = lane; q < ppg; q += 64) s += pp[q]`
            for (int q = lane + 256; q < a.ppg; q += 64) s += pp[q];
            gn = wave_sum_f32(s) / (float)a.rows_per_group;
        }
    }
    f32x4 xf[PT];
    if (i > 0) {
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            float mx[9];
            pc_update_row(xv[p], gr[p], zz1[p], zz2[p], gn, gdiff, dt, sqdt, mx);
            if (row[p] < a.nrows && g == 0) {
                const int r = row[p];
                if (a.traj) {
                    float *tr = a.traj + ((size_t)(i - 1) * a.nrows + r) * 9;
#pragma unroll
                    for (int j = 0; j < 6; ++j) tr[j] = xv[p][j];
#pragma unroll
                    for (int j = 0; j < 3; ++j) tr[6 + j] = xv[p][6 + j] + cen[p][j];
                }
#pragma unroll
                for (int j = 0; j < 9; ++j) a.x[(size_t)r * 9 + j] = xv[p][j];
                if (i == a.nsteps) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) mx[6 + j] += cen[p][j];
                    normalize_rot6(mx);
#pragma unroll
                    for (int j = 0; j < 9; ++j) a.mean_x[(size_t)r * 9 + j] = mx[j];
                }
            }
        }
        if (i == a.nsteps) return;
    }
    __builtin_amdgcn_sched_barrier(0);
    gp_chain::begin_deposit<PT>(sg, lds);
#pragma unroll
    for (int p = 0; p < PT; ++p) xf[p] = gp_chain::pose_fragment(xv[p], g);
    float f[PT][POSE];
    float gx9[PT][POSE];  // MODEL 1: J_f^T u, every lane of a row holds all nine components
    if constexpr (MODEL == 0) {
        gp_chain::run<PT>(st, lds, net, xf, f);
    } else {
        float u[PT][POSE];
        f32x4 gx[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int j = 0; j < POSE; ++j) u[p][j] = xv[p][j] / sigma;
        gp_chain::store_cotangent<PT>(lds, u);
        gp_chain::run_vjp<PT>(st, lds, net, xf, f, gx);
        // lane (row, g) holds components 4g .. 4g+3: hand every lane of the row all nine
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int j = 0; j < POSE; ++j) gx9[p][j] = __shfl(gx[p][j & 3], pt + 16 * (j >> 2), 64);
    }
    float nsum = 0.f;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        float q = 0.f, sc9[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            sc9[j] = MODEL == 0 ? f[p][j] / (sigma + 1e-7f) : f[p][j] / sigma + gx9[p][j];
            q += sc9[j] * sc9[j];
        }
        if (row[p] < a.nrows && g == 0) {
#pragma unroll
            for (int j = 0; j < 9; ++j) a.score[(size_t)row[p] * 9 + j] = sc9[j];
            nsum += sqrtf(q);
        }
    }
    nsum = wave_sum_f32(nsum);
    if (lane == 0) a.partials[(size_t)i * a.nparts + (size_t)blockIdx.x * gp_chain::NW + wave] = nsum;
}

template <typename K>
int set_lds(K kern, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? GP_OK : GP_ELAUNCH;
}

}  // namespace

template <int PT>
static int launch_eval_chain(int R, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x, const float *sigma_dev, int mode,
                             float *out, hipStream_t st) {
    using C = gp_chain::Cfg<PT>;
    static bool attr_done = false;
    if (!attr_done) {
        if (set_lds(score_eval_chain_kernel<PT>, C::LDS_BYTES)) return GP_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((score_eval_chain_kernel<PT>), dim3((R + C::ROWS - 1) / C::ROWS), dim3(gp_chain::NT), C::LDS_BYTES, st, R, k, *net, cvec, tvec, x,
                       sigma_dev, mode, out);
    return gp_launch_status();
}

template <int PT, int MODEL>
static int launch_pc_chain(const PcArgs &a, const gp_scorenet *net, int nwg, hipStream_t st) {
    const size_t lds = MODEL == 0 ? gp_chain::Cfg<PT>::LDS_BYTES : gp_chain::CfgV<PT>::LDS_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        if (set_lds(pc_step_chain_kernel<PT, MODEL>, lds)) return GP_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((pc_step_chain_kernel<PT, MODEL>), dim3(nwg), dim3(gp_chain::NT), lds, st, a, *net);
    return gp_launch_status();
}

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

int gp_score_eval_plan(int tile, int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x,
                       const float *sigma_dev, int mode, float *out, gp_stream_t s) {
    if (nclouds < 0 || k <= 0 || !net || !cvec || !tvec || !x || !sigma_dev || !out || (mode != 0 && mode != 1)) return GP_EINVAL;
    const int R = nclouds * k;
    if (R == 0) return GP_OK;
    const int P = tile == 0 ? score_plan_rows(R, 0, k) : tile;
    hipStream_t st = (hipStream_t)s;
    if (P == 128) return gp_chain::Cfg<2>::fits(k) ? launch_eval_chain<2>(R, k, net, cvec, tvec, x, sigma_dev, mode, out, st) : GP_EINVAL;
    if (P != 16 && P != 32 && P != 64) return GP_EINVAL;
    static bool attr_done = false;
    if (!attr_done) {
        if (set_lds(score_eval_kernel<16>, trunk_lds_bytes<16>()) || set_lds(score_eval_kernel<32>, trunk_lds_bytes<32>()) ||
            set_lds(score_eval_kernel<64>, trunk_lds_bytes<64>()))
            return GP_ELAUNCH;
        attr_done = true;
    }
    if (P == 64)
        hipLaunchKernelGGL(score_eval_kernel<64>, dim3((R + 63) / 64), dim3(TrunkCfg<64>::NT), trunk_lds_bytes<64>(), st, R, k, *net, cvec, tvec, x, sigma_dev,
                           mode, out);
    else if (P == 16)
        hipLaunchKernelGGL(score_eval_kernel<16>, dim3((R + 15) / 16), dim3(TrunkCfg<16>::NT), trunk_lds_bytes<16>(), st, R, k, *net, cvec, tvec, x, sigma_dev,
                           mode, out);
    else
        hipLaunchKernelGGL(score_eval_kernel<32>, dim3((R + 31) / 32), dim3(TrunkCfg<32>::NT), trunk_lds_bytes<32>(), st, R, k, *net, cvec, tvec, x, sigma_dev,
                           mode, out);
    return gp_launch_status();
}

int gp_score_eval(int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x, const float *sigma_dev, int mode,
                  float *out, gp_stream_t s) {
    return gp_score_eval_plan(0, nclouds, k, net, cvec, tvec, x, sigma_dev, mode, out, s);
}

int gp_pc_tile_rows(int ngroups, int nclouds_per_group, int k) {
    if (ngroups <= 0 || nclouds_per_group < 0 || k <= 0) return GP_EINVAL;
    const int rg = nclouds_per_group * k;
    int P = score_tile_rows(ngroups * rg);
    if (ngroups > 1 && rg % P != 0) P = 16;  // tiles must not straddle groups
    if (ngroups > 1 && rg % P != 0) return GP_EINVAL;
    return P;
}

// rows per partial sum of |score| for a plan
static int pc_rows_per_partial(int P) { return P <= 64 ? P : P / gp_chain::NW; }  // tile form: per workgroup; chain form: per wave
static int pc_rows_per_wg(int P) { return P; }

int gp_pc_layout(int model, int tile, int ngroups, int nclouds_per_group, int k, int *tile_out, int *nparts_out) {
    if ((model != 0 && model != 1) || ngroups <= 0 || nclouds_per_group < 0 || k <= 0 || !tile_out || !nparts_out) return GP_EINVAL;
    const int rg = nclouds_per_group * k;
    int P = tile;
    if (model == 1) {  // the energy model's score needs the backward pass: 16-row tiles (score_bwd.h) or the 128-row chain form (trunk_chain_vjp.h)
        if (P == 0) P = score_plan_rows_vjp(ngroups * rg, ngroups > 1 ? rg : 0, k);
        if (P != 16 && P != 128) return GP_EINVAL;
    }
    if (P == 0) P = score_plan_latency(ngroups * rg, ngroups > 1 ? rg : 0, k);
    if (P == (16 | GP_PLAN_HEADSPLIT)) {  // three workgroups per 16-row tile, one head each: one partial per row and head (PcArgs)
        if (model != 0 || (ngroups > 1 && rg % 16 != 0)) return GP_EINVAL;
        *tile_out = P;
        *nparts_out = 21 * ngroups * rg;  // per step: row-and-head sums of squares [3R], score [9R], state [9R] (PcArgs)
        return GP_OK;
    }
    if (P != 16 && P != 32 && P != 64 && P != 128) return GP_EINVAL;
    if (P == 128 && !gp_chain::Cfg<2>::fits(k)) return GP_EINVAL;
    if (ngroups > 1 && rg % pc_rows_per_wg(P) != 0) return GP_EINVAL;  // a workgroup must not straddle two batches
    const int rpp = pc_rows_per_partial(P);
    *tile_out = P;
    *nparts_out = ngroups * ((rg + pc_rows_per_wg(P) - 1) / pc_rows_per_wg(P)) * (pc_rows_per_wg(P) / rpp);
    return GP_OK;
}

int gp_pc_step_plan(int model, int tile, int ngroups, int nclouds_per_group, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec,
                    const float *tvec_all, const float *sched, const float *z_langevin, const float *z_predictor, const float *centre, float *x,
                    float *mean_x, float *score, float *partials, float *traj, const float *gn_ext, int gn_rows_total, gp_stream_t s) {
    if (ngroups <= 0 || nclouds_per_group < 0 || k <= 0 || step < 0 || step > nsteps || !net || !cvec || !tvec_all || !sched || !z_langevin ||
        !z_predictor || !centre || !x || !mean_x || !score || !partials || gn_rows_total < 0)
        return GP_EINVAL;
    const int rg = nclouds_per_group * k, R = ngroups * rg;
    if (R == 0) return GP_OK;
    int P = 0, nparts = 0;
    const int rc = gp_pc_layout(model, tile, ngroups, nclouds_per_group, k, &P, &nparts);
    if (rc != GP_OK) return rc;
    if (model == 1 && (!net->w_headx_t || !net->w_pose2_t || !net->w_pose0_t)) return GP_EINVAL;
    const bool split = P == (16 | GP_PLAN_HEADSPLIT);
    if (split) {
        if (gn_ext) return GP_EINVAL;  // a sharded batch's callers sum whole-tile partials between the launches: tile plans only
        P = 16;
    }
    PcArgs a;
    a.nrows = R, a.kcand = k, a.step = step, a.nsteps = nsteps;
    a.nparts = nparts, a.ppg = nparts / ngroups, a.rows_per_group = rg;
    a.wgpg = (rg + pc_rows_per_wg(P) - 1) / pc_rows_per_wg(P);
    a.cvec = cvec, a.tvec_all = tvec_all, a.sched = sched, a.z_lang = z_langevin, a.z_pred = z_predictor, a.centre = centre;
    a.x = x, a.mean_x = mean_x, a.score = score, a.partials = partials, a.traj = traj;
    a.gn_ext = gn_ext, a.ngroups = ngroups, a.gn_rows = (float)gn_rows_total;
    hipStream_t st = (hipStream_t)s;
    const int nwg = a.wgpg * ngroups;
    if (P == 128) return model == 1 ? launch_pc_chain<2, 1>(a, net, nwg, st) : launch_pc_chain<2, 0>(a, net, nwg, st);
    static bool attr_done = false;
    if (!attr_done) {
        if (set_lds(pc_step_kernel<16, 0>, trunk_lds_bytes<16>()) || set_lds(pc_step_kernel<16, 0, true>, trunk_lds_bytes<16>()) ||
            set_lds(pc_step_kernel<32, 0>, trunk_lds_bytes<32>()) ||
            set_lds(pc_step_kernel<64, 0>, trunk_lds_bytes<64>()) ||
            set_lds(pc_step_kernel<16, 1>, gp_bwd::LDS_BYTES))
            return GP_ELAUNCH;
        attr_done = true;
    }
    if (model == 1)
        hipLaunchKernelGGL((pc_step_kernel<16, 1>), dim3(nwg), dim3(TrunkCfg<16>::NT), gp_bwd::LDS_BYTES, st, a, *net);
    else if (split)
        hipLaunchKernelGGL((pc_step_kernel<16, 0, true>), dim3(3 * nwg), dim3(TrunkCfg<16>::NT), trunk_lds_bytes<16>(), st, a, *net);
    else if (P == 16)
        hipLaunchKernelGGL((pc_step_kernel<16, 0>), dim3(nwg), dim3(TrunkCfg<16>::NT), trunk_lds_bytes<16>(), st, a, *net);
    else if (P == 64)
        hipLaunchKernelGGL((pc_step_kernel<64, 0>), dim3(nwg), dim3(TrunkCfg<64>::NT), trunk_lds_bytes<64>(), st, a, *net);
    else
        hipLaunchKernelGGL((pc_step_kernel<32, 0>), dim3(nwg), dim3(TrunkCfg<32>::NT), trunk_lds_bytes<32>(), st, a, *net);
    return gp_launch_status();
}

int gp_pc_step_coupled(int ngroups, int nclouds_per_group, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec,
                       const float *tvec_all, const float *sched, const float *z_langevin, const float *z_predictor, const float *centre,
                       float *x, float *mean_x, float *score, float *partials, float *traj, const float *gn_ext, gp_stream_t s) {
    // the entry points that predate gp_pc_layout stay in the TILE form: `partials` keeps the size their contract states,
    // nsteps x ngroups x ceil(rows per group / gp_pc_tile_rows) - the chain form writes one partial per wave and is reached through
    // gp_pc_layout + gp_pc_step_plan only
    const int legacy_tile = gp_pc_tile_rows(ngroups, nclouds_per_group, k);
    if (legacy_tile < 0) return legacy_tile;
    return gp_pc_step_plan(0, legacy_tile, ngroups, nclouds_per_group, k, step, nsteps, net, cvec, tvec_all, sched, z_langevin, z_predictor, centre, x, mean_x, score,
                           partials, traj, gn_ext, 0, s);
}

int gp_pc_step_grouped(int ngroups, int nclouds_per_group, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec,
                       const float *tvec_all, const float *sched, const float *z_langevin, const float *z_predictor, const float *centre,
                       float *x, float *mean_x, float *score, float *partials, float *traj, gp_stream_t s) {
    const int legacy_tile = gp_pc_tile_rows(ngroups, nclouds_per_group, k);
    if (legacy_tile < 0) return legacy_tile;
    return gp_pc_step_plan(0, legacy_tile, ngroups, nclouds_per_group, k, step, nsteps, net, cvec, tvec_all, sched, z_langevin, z_predictor, centre, x, mean_x,
                           score, partials, traj, nullptr, 0, s);
}

int gp_pc_step(int nclouds, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec, const float *tvec_all,
               const float *sched, const float *z_langevin, const float *z_predictor, const float *centre, float *x, float *mean_x,
               float *score, float *partials, float *traj, gp_stream_t s) {
    if (nclouds < 0 || k <= 0) return GP_EINVAL;
    return gp_pc_step_plan(0, score_tile_rows(nclouds * k), 1, nclouds, k, step, nsteps, net, cvec, tvec_all, sched, z_langevin, z_predictor, centre, x, mean_x,
                           score, partials, traj, nullptr, 0, s);
}

}  // extern "C"
