// OPT-IN, EXPLORATORY (round 5): one launch of the predictor-corrector sampler (cond_pc_sampler, samplers.py:102-160) with the score network's
// dense layers on the BF16 matrix pipe as three-term split products, fp32 accumulation (bf16x3.h) - the arithmetic of sa_bf16x3.hip applied to
// pc_step_chain_kernel's job.  NOT the default and in no parity claim: the headline numbers are the fp32 kernels' (scorenet.hip, trunk_chain.h).
// PC sampler only: the adaptive RK45 driver keeps the fp32 trunk (its error estimate compares differences of right-hand sides).
//
// Form: 8 waves per workgroup, 16 rows per wave = 128 rows per workgroup (the fp32 chain form's row block, so the same batches-per-launch
// rules apply), one workgroup per CU.  A wave carries its rows from the sampler update to the score in registers (the D fragment of a layer
// IS the next layer's operand, two chunks per k-block of v_mfma_f32_16x16x32_bf16); ALL weights stream through a 3-slot LDS ring in 33 slices
// of 32 KB = (one 32-wide k-block) x (16 output chunks) x (hi, lo): pose_encoder.0 (1), pose_encoder.2 (8), three heads (8 each); one barrier
// per slice.  The three Linear(256, 3) output layers are fp32 dot products on the VALU, taken on the accumulator fragments.
// Bound: the LDS fragment reads (every wave reads every slice: 8 MB per workgroup) and the barriers, not the matrix pipe.
#include "bf16x3.h"
#include "score_trunk.h"

namespace {

using namespace gp_trunk;
using namespace gp_bf16x3;

struct PcBfArgs {
    int nrows, kcand, step, nsteps;
    int nparts, ppg, rows_per_group, wgpg;  // one partial sum of |score| per WAVE: nparts = workgroups x 8
    const float *cvec, *tvec_all, *sched, *z_lang, *z_pred, *centre;
    float *x, *mean_x, *score, *partials, *traj;
    const bf16x8 *w0;   // pose_encoder.0   [1][16][2][64]   k = component index (natural order, zero padded to 32)
    const bf16x8 *w2;   // pose_encoder.2   [8][16][2][64]   k order of the register chain (weights.pack_bf16x3)
    const bf16x8 *wh;   // stacked heads    [8][48][2][64]
    const float *b0, *b2, *w_out, *b_out;  // fp32: biases [256], [256]; output layers [9][256], [9]
};

constexpr int BF_NW = 8, BF_NT = 512, BF_ROWS = 128, BF_NCL = 4;
constexpr int BF_SLICE = 16 * 2 * 64;  // bf16x8 (16 B) per slice = 32 KB
constexpr int BF_PER_T = BF_SLICE / BF_NT;
constexpr int BF_NSLICES = 33;
// LDS: ring [3][SLICE] bf16x8 | w_out [9][256] | b0 [256] | b2 [256] | cvt [NCL][768] = cvec[cloud] + tvec   (floats)
constexpr int BF_OFF_WOUT = 3 * BF_SLICE * 4, BF_OFF_B0 = BF_OFF_WOUT + POSE * HID, BF_OFF_B2 = BF_OFF_B0 + HID, BF_OFF_CVT = BF_OFF_B2 + HID,
              BF_TOTAL = BF_OFF_CVT + BF_NCL * HEADS;
constexpr size_t BF_LDS_BYTES = (size_t)BF_TOTAL * sizeof(float);

__device__ __forceinline__ const bf16x8 *bf_slice(const PcBfArgs &a, int s) {
    s = s < BF_NSLICES ? s : BF_NSLICES - 1;  // the ring runs ahead: requests past the end re-read the last slice (never used)
    if (s == 0) return a.w0;
    if (s <= 8) return a.w2 + (size_t)(s - 1) * BF_SLICE;
    const int h = (s - 9) >> 3, kb = (s - 9) & 7;
    return a.wh + ((size_t)kb * 48 + 16 * h) * 2 * 64;
}

__global__ __launch_bounds__(BF_NT) void pc_step_bf16x3_kernel(PcBfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    bf16x8 *ring = reinterpret_cast<bf16x8 *>(lds);
    float *woutl = lds + BF_OFF_WOUT, *b0l = lds + BF_OFF_B0, *b2l = lds + BF_OFF_B2, *cvtl = lds + BF_OFF_CVT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pt = lane & 15, g = lane >> 4, i = a.step;
    const int wg_row0 = blockIdx.x * BF_ROWS;
    const int row = wg_row0 + wave * 16 + pt;
    const int r = row < a.nrows ? row : a.nrows - 1;  // rows past the end: clamped duplicates (computed, never stored)
    // ---- the row's operands first, the ring prologue and the staged epilogue operands behind them
    float xv[9], gr[9], zz1[9], zz2[9], cen[3] = {0.f, 0.f, 0.f};
    float gdiff = 0.f, dt = 0.f, sqdt = 0.f, gn = 1.f, sigma = 1.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) xv[j] = a.x[(size_t)r * 9 + j];
    if (i > 0) {
        const float *z1 = a.z_lang + ((size_t)(i - 1) * a.nrows + r) * 9;
        const float *z2 = a.z_pred + ((size_t)(i - 1) * a.nrows + r) * 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            gr[j] = a.score[(size_t)r * 9 + j];
            zz1[j] = z1[j];
            zz2[j] = z2[j];
        }
        const float *cp = a.centre + (size_t)(r / a.kcand) * 3;
        cen[0] = cp[0], cen[1] = cp[1], cen[2] = cp[2];
        const float *sc = a.sched + (size_t)(i - 1) * 4;
        gdiff = sc[1], dt = sc[2], sqdt = sc[3];
        // the batch mean of |score_{i-1}|: every wave reduces its batch's partial sums in the same fixed order
        const float *pp = a.partials + (size_t)(i - 1) * a.nparts + (size_t)(blockIdx.x / a.wgpg) * a.ppg;
        float s = 0.f;
        for (int q = lane; q < a.ppg; q += 64) s += pp[q];
        gn = wave_sum_f32(s) / (float)a.rows_per_group;
    }
    if (i < a.nsteps) {
        sigma = a.sched[(size_t)i * 4 + 0];
        // ring prologue: slices 0 and 1 into slots 0 and 1
#pragma unroll
        for (int u = 0; u < BF_PER_T; ++u) {
            ring[0 * BF_SLICE + tid + u * BF_NT] = bf_slice(a, 0)[tid + u * BF_NT];
            ring[1 * BF_SLICE + tid + u * BF_NT] = bf_slice(a, 1)[tid + u * BF_NT];
        }
        for (int e = tid; e < POSE * HID; e += BF_NT) woutl[e] = a.w_out[e];
        for (int e = tid; e < HID; e += BF_NT) b0l[e] = a.b0[e], b2l[e] = a.b2[e];
        const float *tvec = a.tvec_all + (size_t)i * HEADS;
        const int cloud0 = wg_row0 / a.kcand, nclouds = (a.nrows + a.kcand - 1) / a.kcand;
        for (int e = tid; e < BF_NCL * HEADS; e += BF_NT) {
            const int c = e / HEADS, o = e - c * HEADS;
            const int cl = cloud0 + c < nclouds ? cloud0 + c : nclouds - 1;
            cvtl[e] = a.cvec[(size_t)cl * HEADS + o] + tvec[o];
        }
    }
    if (i > 0) {
        float mx[9];
        pc_update_row(xv, gr, zz1, zz2, gn, gdiff, dt, sqdt, mx);
        if (row < a.nrows && g == 0) {
            if (a.traj) {
                float *tr = a.traj + ((size_t)(i - 1) * a.nrows + row) * 9;
#pragma unroll
                for (int j = 0; j < 6; ++j) tr[j] = xv[j];
#pragma unroll
                for (int j = 0; j < 3; ++j) tr[6 + j] = xv[6 + j] + cen[j];
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) a.x[(size_t)row * 9 + j] = xv[j];
            if (i == a.nsteps) {
#pragma unroll
                for (int j = 0; j < 3; ++j) mx[6 + j] += cen[j];
                normalize_rot6(mx);
#pragma unroll
                for (int j = 0; j < 9; ++j) a.mean_x[(size_t)row * 9 + j] = mx[j];
            }
        }
        if (i == a.nsteps) return;
    }
    // slice 2 travels in registers until step 0 deposits it
    bf16x8 hold[BF_PER_T];
#pragma unroll
    for (int u = 0; u < BF_PER_T; ++u) hold[u] = bf_slice(a, 2)[tid + u * BF_NT];
    __syncthreads();
    int gstep = 0;
    // one ring step: slice gstep + 2 (held since the previous step) -> its slot (last read in step gstep - 1), request slice gstep + 3;
    // acc[nc] += W[slice gstep][nc] . (xh, xl) for the 16 output chunks, two at a time; one barrier
    auto ring_step = [&](f32x4 (&acc)[16], const bf16x8 xh, const bf16x8 xl) {
        {
            bf16x8 *dst = ring + ((gstep + 2) % 3) * BF_SLICE;
#pragma unroll
            for (int u = 0; u < BF_PER_T; ++u) dst[tid + u * BF_NT] = hold[u];
            const bf16x8 *src = bf_slice(a, gstep + 3);
#pragma unroll
            for (int u = 0; u < BF_PER_T; ++u) hold[u] = src[tid + u * BF_NT];
        }
        const bf16x8 *slot = ring + (gstep % 3) * BF_SLICE;
#pragma unroll
        for (int n0 = 0; n0 < 16; n0 += 2) {
            bf16x8 wh[2], wl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                wh[u] = slot[((n0 + u) * 2 + 0) * 64 + lane];
                wl[u] = slot[((n0 + u) * 2 + 1) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[u], xh, acc[n0 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[u], xh, acc[n0 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[u], xl, acc[n0 + u], 0, 0, 0);
        }
        ++gstep;
        __syncthreads();
    };
    f32x4 acc[16];
    auto zero_acc = [&]() {
#pragma unroll
        for (int n = 0; n < 16; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // bias + ReLU + split of a 256-wide hidden layer: k-block m of the next layer = chunks 2m, 2m+1
    auto hidden = [&](const float *bias, bf16x8 (&hh)[8], bf16x8 (&hl)[8]) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const f32x4 v0 = relu4(acc[2 * m] + *reinterpret_cast<const f32x4 *>(bias + 16 * (2 * m) + 4 * g));
            const f32x4 v1 = relu4(acc[2 * m + 1] + *reinterpret_cast<const f32x4 *>(bias + 16 * (2 * m + 1) + 4 * g));
            split8(v0, v1, hh[m], hl[m]);
        }
    };
    // ---- pose_encoder.0: the row's nine components as the one (zero-padded) k-block, natural k order: lane group g holds k = 8g .. 8g+7
    bf16x8 xh, xl;
    {
        f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
        if (g == 0) pa = f32x4{xv[0], xv[1], xv[2], xv[3]}, pb = f32x4{xv[4], xv[5], xv[6], xv[7]};
        if (g == 1) pa = f32x4{xv[8], 0.f, 0.f, 0.f};
        split8(pa, pb, xh, xl);
    }
    bf16x8 h1h[8], h1l[8], h2h[8], h2l[8];
    zero_acc();
    ring_step(acc, xh, xl);
    hidden(b0l, h1h, h1l);
    // ---- pose_encoder.2
    zero_acc();
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) ring_step(acc, h1h[kb], h1l[kb]);
    hidden(b2l, h2h, h2l);
    // ---- the three heads; their Linear(256, 3) output layers as fp32 dot products on the accumulator fragments
    const int cl = r / a.kcand - wg_row0 / a.kcand;  // < NCL (gp_pc_layout admits k only when a workgroup's rows span <= NCL clouds)
    float out9[9];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        zero_acc();
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) ring_step(acc, h2h[kb], h2l[kb]);
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int ch = 16 * n + 4 * g;
            const f32x4 v = relu4(acc[n] + *reinterpret_cast<const f32x4 *>(cvtl + cl * HEADS + 256 * h + ch));
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(woutl + (3 * h + 0) * HID + ch);
            const f32x4 w1 = *reinterpret_cast<const f32x4 *>(woutl + (3 * h + 1) * HID + ch);
            const f32x4 w2 = *reinterpret_cast<const f32x4 *>(woutl + (3 * h + 2) * HID + ch);
            o0 += v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
            o1 += v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
            o2 += v.x * w2.x + v.y * w2.y + v.z * w2.z + v.w * w2.w;
        }
        out9[3 * h + 0] = lane_groups_sum(o0);  // the four lane groups hold the four channel quarters: fixed order, every lane gets the sum
        out9[3 * h + 1] = lane_groups_sum(o1);
        out9[3 * h + 2] = lane_groups_sum(o2);
    }
    float q = 0.f, sc9[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        sc9[j] = (out9[j] + a.b_out[j]) / (sigma + 1e-7f);
        q += sc9[j] * sc9[j];
    }
    float nsum = 0.f;
    if (row < a.nrows && g == 0) {
#pragma unroll
        for (int j = 0; j < 9; ++j) a.score[(size_t)row * 9 + j] = sc9[j];
        nsum = sqrtf(q);
    }
    nsum = wave_sum_f32(nsum);
    if (lane == 0) a.partials[(size_t)i * a.nparts + (size_t)blockIdx.x * BF_NW + wave] = nsum;
}

}  // namespace

extern "C" {

/* rows per workgroup / partial sums per step of the split-bf16 PC launch for (ngroups x nclouds_per_group clouds x k candidates); GP_EINVAL when a
 * workgroup would straddle two batches or its 128 rows could span more than four clouds (k < 43) */
int gp_pc_layout_bf16x3(int ngroups, int nclouds_per_group, int k, int *nparts_out) {
    if (ngroups <= 0 || nclouds_per_group <= 0 || k <= 0 || !nparts_out) return GP_EINVAL;
    const int rg = nclouds_per_group * k;
    if (ngroups > 1 && rg % BF_ROWS != 0) return GP_EINVAL;
    if ((BF_ROWS - 2 + k) / k + 1 > BF_NCL) return GP_EINVAL;
    *nparts_out = ngroups * ((rg + BF_ROWS - 1) / BF_ROWS) * BF_NW;
    return GP_OK;
}

int gp_pc_step_bf16x3(int ngroups, int nclouds_per_group, int k, int step, int nsteps, const float *cvec, const float *tvec_all, const float *sched,
                      const float *z_langevin, const float *z_predictor, const float *centre, float *x, float *mean_x, float *score, float *partials,
                      float *traj, const void *w_pose0_split, const void *w_pose2_split, const void *w_headx_split, const float *b_pose0, const float *b_pose2,
                      const float *w_out, const float *b_out, gp_stream_t s) {
    if (step < 0 || step > nsteps || !cvec || !tvec_all || !sched || !z_langevin || !z_predictor || !centre || !x || !mean_x || !score || !partials ||
        !w_pose0_split || !w_pose2_split || !w_headx_split || !b_pose0 || !b_pose2 || !w_out || !b_out)
        return GP_EINVAL;
    int nparts = 0;
    const int rc = gp_pc_layout_bf16x3(ngroups, nclouds_per_group, k, &nparts);
    if (rc != GP_OK) return rc;
    const int rg = nclouds_per_group * k;
    PcBfArgs a;
    a.nrows = ngroups * rg, a.kcand = k, a.step = step, a.nsteps = nsteps;
    a.wgpg = (rg + BF_ROWS - 1) / BF_ROWS, a.nparts = nparts, a.ppg = a.wgpg * BF_NW, a.rows_per_group = rg;
    a.cvec = cvec, a.tvec_all = tvec_all, a.sched = sched, a.z_lang = z_langevin, a.z_pred = z_predictor, a.centre = centre;
    a.x = x, a.mean_x = mean_x, a.score = score, a.partials = partials, a.traj = traj;
    a.w0 = reinterpret_cast<const bf16x8 *>(w_pose0_split), a.w2 = reinterpret_cast<const bf16x8 *>(w_pose2_split),
    a.wh = reinterpret_cast<const bf16x8 *>(w_headx_split);
    a.b0 = b_pose0, a.b2 = b_pose2, a.w_out = w_out, a.b_out = b_out;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(pc_step_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS_BYTES) !=
            hipSuccess)
            return GP_ELAUNCH;
        done = true;
    }
    hipLaunchKernelGGL(pc_step_bf16x3_kernel, dim3(a.wgpg * ngroups), dim3(BF_NT), BF_LDS_BYTES, (hipStream_t)s, a);
    return gp_launch_status();
}

}  // extern "C"
