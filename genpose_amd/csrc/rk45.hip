// Probability-flow ODE sampler on the device: Dormand-Prince 5(4) with scipy's step controller, f64 state,
// f32 score network - the semantics of cond_ode_sampler (networks/gf_algorithms/samplers.py:163-227) driving
// scipy.integrate.solve_ivp(method='RK45', rtol=atol=1e-5) (scipy/_ivp/rk.py, common.py).
//
// The reference keeps the f64 state on the HOST and crosses PCIe twice per function evaluation
// (samplers.py:187,191).  Here state, stage derivatives, t, h and the accept/reject decision all live in HBM:
//   rk45_init_a / _b / _c : f0, Hairer initial step (two evaluations), first stage times
//   per attempt           : time_embed(6 stage times) -> 6 fused stage kernels (stage update + score) -> rk45_decide
//   rk45_finish           : denoise step (samplers.py:209-218), normalize_rotation, + centre
// The error norm is the reference's batch-global RMS over all R*9 components: per-tile partial sums, reduced in
// a fixed order by the single-workgroup decide kernel (deterministic).
#include "score_bwd.h"
#include "trunk_chain_vjp.h"

namespace {

using namespace gp_trunk;

// Dormand-Prince tableau (Dormand & Prince 1980; the constants scipy's RK45 uses)
__constant__ double DP_C[7] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
__constant__ double DP_A[7][6] = {
    {0, 0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
    {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84},  // B (stage 6 = y_new)
};
__constant__ double DP_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};

constexpr double SAFETY = 0.9, MIN_FACTOR = 0.2, MAX_FACTOR = 10.0;
constexpr double SIG_MIN = 0.01, SIG_RATIO = 50.0 / 0.01;

// Device-resident solver state (one per sampler instance).
struct Rk45State {
    double t, h_abs, t_bound, direction, rtol, atol;
    double h;              // signed step of the attempt in flight
    double t_new;          // its end point (exactly t_bound on the last step - rk.py keeps t_new, not t + h)
    double d0, d1, h0;     // initial-step scratch
    double err_norm;       // of the last attempt
    int status;            // 0 running, 1 finished, -1 step size too small
    int last_accepted;     // previous attempt accepted -> stage 1 commits y_new/K6 first
    int step_rejected;     // a rejection happened since the last accepted step
    int n_attempts, n_accepted, nfev;
    int traj_cap;
    // dense output (solve_ivp with t_eval): emit the t_eval points passed by the accepted step [emit_begin, emit_end)
    int n_eval, next_eval, emit_begin, emit_end;
    double t_old, h_acc;    // start and signed size of the last accepted step (st->h is already the NEXT attempt's)
    const double *t_eval;   // device array [n_eval]
    double Pm[7][4];        // RK45 dense-output matrix P (passed from the host, scipy's published constants)
    float stage_t[8];      // f32 times fed to the network (stages 1..6 at [1..6]; [0] = misc evaluations)
    float stage_sigma[8];  // f32 sigma(t) the network divides by (scorenet.py:205,217)
    double stage_g2[8];    // f64 g(t)^2 of the PF-ODE right-hand side (sde.py:20-24 on a 0-dim f64 tensor)
    double log_t[512], log_h[512], log_err[512];  // per attempt (diagnostics / parity tests)
    int log_acc[512];
    // shared-chunk plan (GP_PLAN_SHARED): progress word of every shared 16-row chunk, 8 * attempt + stages completed (appended: the
    // offsets gp_rk45_state_layout exports do not move)
    int xflags[64];
    int xfail;  // a bounded wait of the shared-chunk plan ran out: the controller ends the solve with status -2
};

__device__ __forceinline__ void set_stage(Rk45State *st, int slot, double t) {
    const float tf = (float)t;  // torch.ones(R,1) * t  -> f32 (samplers.py:192)
    st->stage_t[slot] = tf;
    st->stage_sigma[slot] = 0.01f * powf(5000.0f, tf);
    const double sg = SIG_MIN * pow(SIG_RATIO, t);
    const double g = sg * sqrt(2.0 * (log(50.0) - log(0.01)));
    st->stage_g2[slot] = g * g;
}

struct OdeArgs {
    int nrows, kcand, nblocks;
    int ngroups, bpg, rows_per_group;  // independent batches laid out back to back: one solver state (step control) per group
    // ragged groups (different numbers of clouds per group; null = equal groups): per workgroup {group, first row, end row},
    // per group {first workgroup, workgroups, rows, first row}
    const int *blk_info, *grp_info;
    const float *cvec;
    float *tvec;               // [ngroups][8][768] time embedding per stage slot (slot-indexed like stage_t), written by the controller kernels
    const float *centre;
    Rk45State *st;
    double *y, *ynew, *K;      // y, ynew [R*9]; K [7][R*9]
    double *partials;          // [3][nblocks]
    double *traj;              // [traj_cap][R*9] accepted states (raw, un-normalised) or null
    float *x32;                // [R*9] scratch
    const float *probe;        // [R][9] Hutchinson probe of the likelihood ODE (model 2) or null
    int ncomp;                 // state components per row: 9 (pose), 10 for the likelihood ODE (pose + log-density change)
    // a batch sharded over several GPUs: the norms of the step controller run over ALL its rows.  ext_sums [2][ngroups]: this rank's
    // per-group sums of squares (rk45_group_sums_kernel), all-reduced by the caller between the stage kernels and the controller;
    // ext_rows = rows of a group over all ranks.  null: the controller reduces the local partials itself.
    double *ext_sums;
    int ext_rows;
    // head-split plan (GP_PLAN_HEADSPLIT; score model, 16-row tiles, the latency regime): hsplit = 3 workgroups per tile, workgroup 3 t + h
    // evaluates head h of tile t and owns components 3 h .. 3 h + 2 of its rows; every count above (nblocks, bpg, the ragged tables) stays
    // in TILES, the partial sums are laid out [3][nblocks * hsplit], one per workgroup.  1 = one workgroup per tile.
    int hsplit;
};

// MODEL of the right-hand side (the same Dormand-Prince driver integrates all three):
//   0  probability-flow ODE of the score network           dx/dt = -g^2/2 . f / (sigma + 1e-7)                 (samplers.py:163-227)
//   1  the same ODE driven by the ENERGY network's score    ... . d/dx <x, f(x)/sigma>                          (posenet.py:94-130, energynet.py:200-222)
//   2  likelihood ODE of the score network                  d[x, logp]/dt = -g^2/2 . [score, probe^T J_score probe]   (samplers.py:22-99)
// Models 1 and 2 need the backward pass of the trunk (score_bwd.h) and run on 16-row tiles.
template <int MODEL>
struct OdeModel {
    static constexpr int NC = MODEL == 2 ? 10 : POSE;
    static constexpr bool BWD = MODEL != 0;
};

// which group / rows does this workgroup serve
template <int P>
__device__ __forceinline__ void ode_block(const OdeArgs &a, int tile, int &grp, int &row0, int &row_end) {
    if (a.blk_info) {
        const int *bi = a.blk_info + 3 * tile;
        grp = bi[0], row0 = bi[1], row_end = bi[2];
    } else {
        grp = tile / a.bpg, row0 = tile * P, row_end = a.nrows;
    }
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    const int nw = blockDim.x >> 6;  // 4 or 8 waves; fixed summation order
    double s = sh[0];
    for (int w = 1; w < nw; ++w) s += sh[w];
    __syncthreads();
    return s;
}

// Loads / stores of state the shared-chunk plan hands from one workgroup to another INSIDE a launch (XWG): relaxed atomics at agent scope,
// i.e. sc1 accesses that are served by the memory side of the L2s - no stale line of another XCD's L2 or of a CU's vector cache can
// answer them, and no cache has to be invalidated (an acquire fence at agent scope would drop the XCD's cached weights with it).
template <bool XWG>
__device__ __forceinline__ double ldx(const double *p) {
    if constexpr (XWG)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        return *p;
}
template <bool XWG>
__device__ __forceinline__ void stx(double *p, double v) {
    if constexpr (XWG)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

// Fused stage kernel.  STAGE 1..6: Runge-Kutta stage;  STAGE 0: f0 = fun(t0, y0) (+ d0,d1 partials);
// STAGE 7: f1 = fun(t0 + h0*dir, y0 + h0*dir*f0) (+ d2 partial).
// One stage for the workgroup's tile.  Returns false when the workgroup has nothing to do (padding workgroup, finished solve).
// SPLIT: the head-split plan (OdeArgs::hsplit = 3): this workgroup evaluates ONE head of its tile and owns that head's three state
// components - it computes the stage input of all nine (the network needs them; element-local arithmetic, identical in the three
// workgroups of a tile), and commits / stores / sums only its own.  What a stage reads of earlier stages (K_q of all nine components)
// was written by other workgroups: the stages of an attempt are separate launches under this plan.
// XWG / wg_tile / wg_part (shared-chunk plan): the tile and the partial-sum slot come from the caller instead of blockIdx.x, and XWG marks a
// tile whose state other workgroups of the same launch have written or will read (ldx / stx above).
template <int P, int STAGE, int MODEL, bool SPLIT = false, bool XWG = false, bool ORD8 = false>
__device__ __forceinline__ bool rk45_stage_body(const OdeArgs &a, const gp_scorenet &net, float *lds, double *sh, int wg_tile = -1, int wg_part = -1) {
    static_assert(MODEL == 0 || P == gp_bwd::DP, "the backward pass runs on 16-row tiles");
    static_assert(!SPLIT || (MODEL == 0 && P == 16), "head-split: score model, 16-row tiles");
    using L = TrunkLds<P, OdeModel<MODEL>::BWD>;
    constexpr int NC = OdeModel<MODEL>::NC;
    const int tid = threadIdx.x;
    const int wgi = wg_tile >= 0 ? wg_tile : (int)blockIdx.x, part = wg_part >= 0 ? wg_part : (int)blockIdx.x;
    const int tile = SPLIT ? wgi / 3 : wgi, hsel = SPLIT ? wgi - 3 * tile : 0;
    auto owned = [&](int j) { return !SPLIT || j / 3 == hsel; };
    int grp, row0, rend;
    ode_block<P>(a, tile, grp, row0, rend);
    if (row0 >= rend) return false;  // padding workgroup of a ragged launch (tables sized for a capacity)
    Rk45State *st = a.st + grp;
    if (STAGE >= 1 && STAGE <= 6 && st->status != 0) return false;
    const size_t n = (size_t)a.nrows * NC;
    const int slot = (STAGE >= 1 && STAGE <= 6) ? STAGE : 0;
    const float *tvec = a.tvec + ((size_t)grp * 8 + slot) * HEADS;
    TrunkPre<P> pre;
    trunk_begin<P>(net, pre, a.cvec, tvec, row0, rend, a.kcand);
    const double h = st->h;
    const float sigma = st->stage_sigma[slot];  // requested now, used after the trunk
    const double g2 = st->stage_g2[slot];
    // stage input, one (row, component) element per thread: y (+ h * sum_q a_sq K_q) in f64 -> f32 network input in LDS
    for (int e = tid; e < P * 16; e += TrunkCfg<P>::NT) {
        const int rr = e >> 4, j = e & 15;
        float *xr = lds + rr * L::LD0;
        if (j >= NC) {
            xr[j] = 0.f;
            continue;
        }
        const bool live = row0 + rr < rend;
        const int r = live ? row0 + rr : rend - 1;  // rows past the end: clamped duplicates (computed, never stored)
        const size_t ge = (size_t)r * NC + j;
        const bool commit = STAGE == 1 && st->last_accepted;
        double yv = commit ? ldx<XWG>(a.ynew + ge) : ldx<XWG>(a.y + ge);
        if (commit && live && owned(j)) {
            // commit the previous accepted step for this element (element-local: no other thread touches it)
            stx<XWG>(a.y + ge, yv);
            stx<XWG>(a.K + ge, ldx<XWG>(a.K + 6 * n + ge));
        }
        if (STAGE >= 1 && STAGE <= 6) {
            double dy = 0.0;
#pragma unroll
            for (int q = 0; q < STAGE; ++q) {
                const double kq = (q == 0 && commit) ? ldx<XWG>(a.K + 6 * n + ge) : ldx<XWG>(a.K + (size_t)q * n + ge);
                dy += kq * DP_A[STAGE][q];
            }
            yv = yv + dy * h;  // rk.py: dy = dot(K[:s].T, a[:s]) * h ; y + dy   (stage 6: y + h * dot(K[:-1].T, B))
            if (STAGE == 6 && live && owned(j)) stx<XWG>(a.ynew + ge, yv);
        } else if (STAGE == 7) {
            yv = yv + st->h0 * st->direction * a.K[ge];  // common.py: y1 = y0 + h0 * direction * f0
        }
        xr[j] = j < POSE ? (float)yv : 0.f;  // torch.tensor(x, dtype=float32) (samplers.py:191); the log-density component is no network input
    }
    if (MODEL == 2) gp_bwd::load_probe_tile(lds, a.probe, row0, rend);
    __syncthreads();
    // right-hand side per (row, component): F[r * ldf + j]
    const float *F;
    int ldf;
    if constexpr (MODEL == 0) {
        // (ORD8: the shared-chunk plan's tiles - the output sums in the order of the 32- / 64-row tiles, score_trunk.h)
        trunk_ftheta<P, false, TrunkNoEmit, SPLIT, ORD8 || P == 48>(lds, net, a.cvec, tvec, row0, rend, a.kcand, pre, TrunkNoEmit(), hsel);
        F = lds + L::OFF_H1, ldf = L::LDH;
    } else {
        F = gp_bwd::score_vjp_tile<MODEL == 1 ? gp_bwd::ENERGY : gp_bwd::SCORE_DIV>(lds, net, a.cvec, tvec, row0, rend, a.kcand, pre, sigma);
        ldf = gp_bwd::LDS_OUT;
    }
    double *Kout = a.K + (size_t)(STAGE == 7 ? 1 : (STAGE == 0 ? 0 : (STAGE == 6 ? 6 : STAGE))) * n;
    double acc0 = 0.0, acc1 = 0.0;
    for (int e = tid; e < P * NC; e += TrunkCfg<P>::NT) {
        const int r = e / NC, j = e - r * NC;
        if (row0 + r >= rend || !owned(j)) continue;
        const size_t ge = (size_t)(row0 + r) * NC + j;
        const float rhs = MODEL == 0 ? F[r * ldf + j] / (sigma + 1e-7f) : F[r * ldf + j];  // score component (j = 9: divergence estimate)
        const double kv = 0.0 - (0.5 * g2) * (double)rhs;  // drift - 0.5 * g^2 * score (samplers.py:198; :83-86 for the log-density)
        stx<XWG>(Kout + ge, kv);
        if (STAGE == 0) {
            const double yv = a.y[ge];
            const double sc = st->atol + fabs(yv) * st->rtol;
            acc0 += (yv / sc) * (yv / sc);
            acc1 += (kv / sc) * (kv / sc);
        } else if (STAGE == 7) {
            const double sc = st->atol + fabs(a.y[ge]) * st->rtol;
            const double d = (kv - a.K[ge]) / sc;
            acc0 += d * d;
        } else if (STAGE == 6) {
            double er = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) er += ldx<XWG>(a.K + (size_t)q * n + ge) * DP_E[q];
            er += kv * DP_E[6];
            er *= h;  // rk.py: dot(K.T, E) * h
            const double yo = ldx<XWG>(a.y + ge), yn = ldx<XWG>(a.ynew + ge);
            const double sc = st->atol + fmax(fabs(yo), fabs(yn)) * st->rtol;
            acc0 += (er / sc) * (er / sc);
        }
    }
    if (STAGE == 0 || STAGE == 6 || STAGE == 7) {
        const double s0 = block_sum(acc0, sh);
        if (tid == 0) a.partials[part] = s0;
        if (STAGE == 0) {
            const double s1 = block_sum(acc1, sh);
            if (tid == 0) a.partials[a.nblocks * a.hsplit + part] = s1;
        }
    }
    return true;
}

template <int P, int STAGE, int MODEL, bool SPLIT = false>
__global__ __launch_bounds__(TrunkCfg<P>::NT) void rk45_stage_kernel(OdeArgs a, gp_scorenet net) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double sh[8];
    rk45_stage_body<P, STAGE, MODEL, SPLIT>(a, net, lds, sh);
}

// The six stages of an attempt in ONE launch, for the latency regime (16-row tiles: a tracking frame's solve is ~40 stage launches of
// 17 us each on 16 of the 256 CUs, a configs[0] solve one workgroup - five launch boundaries per attempt are pure overhead there).
// Everything a stage reads from earlier stages of the attempt is row-local: K_q of the tile's own rows, written to global memory by
// this workgroup (by other threads of it: a workgroup-scope barrier separates the stages; none of these lines was read before it was
// written, so no stale copy can sit in the CU's vector cache).  Same arithmetic, same order: bit-identical to the six launches.
template <int P, int MODEL>
__global__ __launch_bounds__(TrunkCfg<P>::NT) void rk45_attempt_kernel(OdeArgs a, gp_scorenet net) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double sh[8];
    if (!rk45_stage_body<P, 1, MODEL>(a, net, lds, sh)) return;  // (wave-uniform: the whole workgroup leaves together)
    __syncthreads();
    rk45_stage_body<P, 2, MODEL>(a, net, lds, sh);
    __syncthreads();
    rk45_stage_body<P, 3, MODEL>(a, net, lds, sh);
    __syncthreads();
    rk45_stage_body<P, 4, MODEL>(a, net, lds, sh);
    __syncthreads();
    rk45_stage_body<P, 5, MODEL>(a, net, lds, sh);
    __syncthreads();
    rk45_stage_body<P, 6, MODEL>(a, net, lds, sh);
}

// The SHARED-CHUNK plan (GP_PLAN_SHARED, round 6; score model, one group): an attempt as ONE launch of one workgroup per CU in which the
// 16-row chunks that do not divide over the CUs are shared ACROSS STAGES instead of costing every stage a fourth round.
//   12 800 rows (scripts/eval_single.sh: 256 clouds x 50 candidates) = 800 chunks on 256 CUs: 3 whole chunks per CU and 32 left over.
//   Per-stage launches are bound by the busiest CU - 4 chunks - whatever the tile shape (72 us per stage on 64-row tiles, 200 CUs busy).
//   Here workgroup w owns rows [w PW, (w + 1) PW) (PW = 16 x whole chunks per CU) for all six stages (row-local, as rk45_attempt_kernel),
//   and the 6 x 32 (chunk, stage) units of the left-over chunks are dealt out one per workgroup: workgroup 6 j + s - 1 evaluates stage s
//   of shared chunk j as a 16-row tile just before its own stage s.  The busiest CU then does 6 x PW rows + ONE 16-row tile per attempt
//   instead of 6 x (PW + 16) rows.
// A shared chunk's stages run on six different CUs inside one launch: its state (y, y_new, K) moves through the memory side of the L2s
// (ldx / stx) and a progress word per chunk orders the stages: 8 x attempt + s once stage s is complete.  The producer of stage s - 1 ran
// it one whole tile pass earlier in its own sequence than the consumer needs it, so the wait is a formality - and it is BOUNDED: a
// consumer that never sees its word fails the solve (status -2) instead of hanging the device.  Workgroups wait only on lower-numbered
// workgroups, which the dispatcher starts first.
__device__ __forceinline__ bool shared_chunk_wait(int *flag, int want, Rk45State *st, double *sh) {
    int *ok = reinterpret_cast<int *>(sh);
    if (threadIdx.x == 0) {
        int it = 0, got = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (got != want && ++it < (1 << 18) && __hip_atomic_load(&st->xfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(4);
            got = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *ok = got == want;
        // (not st->status: the other workgroups read it while they run; the controller kernel turns xfail into status -2)
        if (got != want) __hip_atomic_store(&st->xfail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool r = *ok != 0;
    __syncthreads();
    return r;
}
__device__ __forceinline__ void shared_chunk_signal(int *flag, int value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this thread's stores of the chunk's state have completed at agent scope
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int PW>
__global__ __launch_bounds__(TrunkCfg<PW>::NT) void rk45_attempt_shared_kernel(OdeArgs a, gp_scorenet net, int nshared) {
    static_assert(TrunkCfg<PW>::NT == TrunkCfg<16>::NT, "own tile and shared 16-row tile run on the same waves");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double sh[8];
    Rk45State *st = a.st;
    if (st->status != 0) return;
    const int wg = blockIdx.x, nwg = gridDim.x;
    const int xj = wg / 6, xs = wg - 6 * xj + 1;       // this workgroup's shared unit: stage xs of shared chunk xj
    const bool has_x = xj < nshared;
    const int xtile = (nwg * PW) / 16 + xj;            // the shared chunk as a 16-row tile of the whole batch
    const int xpart = nwg + xj;                        // its slot among the partial sums of the error norm
    const int epoch = 8 * st->n_attempts;              // (written by the controller kernel, between attempts)
    int *flag = st->xflags + xj;
    bool alive = true;
#define GP_SHARED_UNIT(S)                                                                                   \
    if (has_x && xs == S && alive) {                                                                        \
        if (S > 1) alive = shared_chunk_wait(flag, epoch + S - 1, st, sh);                                  \
        if (alive) {                                                                                        \
            rk45_stage_body<16, S, 0, false, true, true>(a, net, lds, sh, xtile, xpart);                          \
            shared_chunk_signal(flag, epoch + S);                                                           \
            __syncthreads();                                                                                \
        }                                                                                                   \
    }
    GP_SHARED_UNIT(1)
    rk45_stage_body<PW, 1, 0, false, false, true>(a, net, lds, sh, wg, wg);
    __syncthreads();
    GP_SHARED_UNIT(2)
    rk45_stage_body<PW, 2, 0, false, false, true>(a, net, lds, sh, wg, wg);
    __syncthreads();
    GP_SHARED_UNIT(3)
    rk45_stage_body<PW, 3, 0, false, false, true>(a, net, lds, sh, wg, wg);
    __syncthreads();
    GP_SHARED_UNIT(4)
    rk45_stage_body<PW, 4, 0, false, false, true>(a, net, lds, sh, wg, wg);
    __syncthreads();
    GP_SHARED_UNIT(5)
    rk45_stage_body<PW, 5, 0, false, false, true>(a, net, lds, sh, wg, wg);
    __syncthreads();
    GP_SHARED_UNIT(6)
    rk45_stage_body<PW, 6, 0, false, false, true>(a, net, lds, sh, wg, wg);
#undef GP_SHARED_UNIT
}

// The same stage in the CHAIN form of the trunk (trunk_chain.h; score model, equal groups, launches of ~32 000 rows and more): a wave
// carries 16 * PT rows from the stage input to K_s in registers.  Lane (row, g) owns components 4g .. 4g+3 of its row - exactly the
// B fragment of the first layer - so the f64 stage arithmetic is spread over all lanes with no exchange.
// MODEL 1 / 2 (the energy model's score, the likelihood ODE): forward + vector-Jacobian chain of trunk_chain_vjp.h.  With ten state
// components per row (model 2) lane group 2 owns components 8 and 9 - the pose's last one and the log-density, which is no network input.
template <int PT, int STAGE, int MODEL>
__global__ __launch_bounds__(gp_chain::NT, 1) void rk45_stage_chain_kernel(OdeArgs a, gp_scorenet net) {
    using C = gp_chain::Cfg<PT>;
    constexpr int NC = OdeModel<MODEL>::NC;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double sh[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pt = lane & 15, g = lane >> 4;
    const int grp = blockIdx.x / a.bpg, wg_row0 = blockIdx.x * C::ROWS;
    Rk45State *st = a.st + grp;
    if (STAGE >= 1 && STAGE <= 6 && st->status != 0) return;
    const size_t n = (size_t)a.nrows * NC;
    const int slot = (STAGE >= 1 && STAGE <= 6) ? STAGE : 0;
    const float *tvec = a.tvec + ((size_t)grp * 8 + slot) * HEADS;
    const double h = st->h;
    const float sigma = st->stage_sigma[slot];
    const double g2 = st->stage_g2[slot];
    const bool commit = STAGE == 1 && st->last_accepted;
    const int nown = g < 2 ? 4 : (g == 2 ? NC - 8 : 0);  // components 4g .. of the row's NC-vector this lane owns
    // ---- stage input: y (+ h * sum_q a_sq K_q) in f64 (requests first, the ring prologue behind them)
    double yv[PT][4];
    size_t ge0[PT];
    bool live[PT];
    float pr[PT][POSE];  // MODEL 2: the row's Hutchinson probe (all nine components in every lane of the row)
    f32x4 prf[PT];       //          and the components 4g .. 4g+3 this lane's gx components meet
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int row = wg_row0 + (wave * PT + p) * 16 + pt;
        live[p] = row < a.nrows;
        const int r = live[p] ? row : a.nrows - 1;  // rows past the end: clamped duplicates (computed, never stored)
        ge0[p] = (size_t)r * NC + 4 * g;
        if constexpr (MODEL == 2) {
#pragma unroll
            for (int j = 0; j < POSE; ++j) pr[p][j] = a.probe[(size_t)r * POSE + j];
            prf[p] = gp_chain::pose_fragment(pr[p], g);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            yv[p][c] = 0.0;
            if (c < nown) {
                const size_t ge = ge0[p] + c;
                double v = commit ? a.ynew[ge] : a.y[ge];
                if (STAGE >= 1 && STAGE <= 6) {
                    double dy = 0.0;
#pragma unroll
                    for (int q = 0; q < STAGE; ++q) {
                        const double kq = (q == 0 && commit) ? a.K[6 * n + ge] : a.K[(size_t)q * n + ge];
                        if (q == 0 && commit && live[p]) {  // commit the previous accepted step for this element
                            a.y[ge] = v;
                            a.K[ge] = kq;
                        }
                        dy += kq * DP_A[STAGE][q];
                    }
                    v = v + dy * h;
                    if (STAGE == 6 && live[p]) a.ynew[ge] = v;
                } else if (STAGE == 7) {
                    v = v + st->h0 * st->direction * a.K[ge];
                }
                yv[p][c] = v;
            }
        }
    }
    gp_chain::State<PT> cs;
    gp_chain::begin<PT>(cs, lds, net, a.cvec, tvec, wg_row0, a.nrows, a.kcand);
    f32x4 xf[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        xf[p] = f32x4{(float)yv[p][0], (float)yv[p][1], (float)yv[p][2], (float)yv[p][3]};
        if (NC > POSE && g == 2) xf[p].y = 0.f;  // the log-density component is no network input
    }
    float f[PT][POSE];
    f32x4 gx[PT];
    float extra[PT];  // MODEL 2: the divergence estimate (J_f^T u) . probe of the row, in every lane
    if constexpr (MODEL == 0) {
        gp_chain::run<PT>(cs, lds, net, xf, f);
    } else {
        float u[PT][POSE];
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int j = 0; j < POSE; ++j) {
                if constexpr (MODEL == 1)
                    u[p][j] = __shfl(xf[p][j & 3], pt + 16 * (j >> 2), 64) / sigma;  // u = x / sigma, x gathered from the lane group that owns it
                else
                    u[p][j] = pr[p][j] / (sigma + 1e-7f);
            }
        gp_chain::store_cotangent<PT>(lds, u);
        gp_chain::run_vjp<PT>(cs, lds, net, xf, f, gx);
        if constexpr (MODEL == 2) {
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                float e = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) e += gx[p][q] * prf[p][q];  // (components beyond 8: both factors are zero)
                e += __shfl_xor(e, 16, 64);
                e += __shfl_xor(e, 32, 64);
                extra[p] = e;
            }
        }
    }
    double *Kout = a.K + (size_t)(STAGE == 7 ? 1 : (STAGE == 0 ? 0 : (STAGE == 6 ? 6 : STAGE))) * n;
    double acc0 = 0.0, acc1 = 0.0;
    f32x4 ff[PT];  // f_theta components 4g .. 4g+3 of the lane's rows
#pragma unroll
    for (int p = 0; p < PT; ++p) ff[p] = gp_chain::pose_fragment(f[p], g);
#pragma unroll
    for (int p = 0; p < PT; ++p) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c >= nown || !live[p]) continue;
            const float fc = ff[p][c];
            const size_t ge = ge0[p] + c;
            float rhs;
            if constexpr (MODEL == 0)
                rhs = fc / (sigma + 1e-7f);
            else if constexpr (MODEL == 1)
                rhs = fc / sigma + gx[p][c];  // d/dx <x, f(x)/sigma> (energynet.py:200-222)
            else
                rhs = (g == 2 && c == 1) ? extra[p] : fc / (sigma + 1e-7f);  // component 9: the divergence estimate (samplers.py:83-86)
            const double kv = 0.0 - (0.5 * g2) * (double)rhs;
            Kout[ge] = kv;
            if (STAGE == 0) {
                const double y0 = a.y[ge];
                const double sc = st->atol + fabs(y0) * st->rtol;
                acc0 += (y0 / sc) * (y0 / sc);
                acc1 += (kv / sc) * (kv / sc);
            } else if (STAGE == 7) {
                const double sc = st->atol + fabs(a.y[ge]) * st->rtol;
                const double d = (kv - a.K[ge]) / sc;
                acc0 += d * d;
            } else if (STAGE == 6) {
                double er = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) er += a.K[(size_t)q * n + ge] * DP_E[q];
                er += kv * DP_E[6];
                er *= h;
                const double yo = a.y[ge], yn = a.ynew[ge];
                const double sc = st->atol + fmax(fabs(yo), fabs(yn)) * st->rtol;
                acc0 += (er / sc) * (er / sc);
            }
        }
    }
    if (STAGE == 0 || STAGE == 6 || STAGE == 7) {
        const double s0 = block_sum(acc0, sh);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = s0;
        if (STAGE == 0) {
            const double s1 = block_sum(acc1, sh);
            if (threadIdx.x == 0) a.partials[a.nblocks + blockIdx.x] = s1;
        }
    }
}

__device__ __forceinline__ double sum_partials(const double *p, int nb, double *sh) {
    double s = 0.0;
    for (int q = threadIdx.x; q < nb; q += 256) s += p[q];
    return block_sum(s, sh);
}

// Time embedding of stage slots [lo, lo + gridDim.x) of every group, straight from the solver states, launched by the phase functions
// right behind the kernel that decided the stage times: tvec[slot] = W1t . relu(Wt1 . [sin(x), cos(x)] + bt1), x = t * W * 2 pi
// (scorenet.py:55-64,111-116) - the arithmetic of time_embed_kernel (scorenet.hip), every output an fmaf chain in k order, so the rows
// equal that kernel's bit for bit.  grid (slots, ngroups, 3 / PER): one slot and PER x 256 of the 768 outputs per workgroup.  With few
// groups the kernel is pure latency: a third of the outputs per workgroup (the hidden layer recomputed by each), a group's six slots
// over 18 CUs, the k loops unrolled so that all their loads are in flight at once; with many groups (tracking: one per sequence)
// it is throughput-bound and one workgroup per slot does all 768 (fused into the one-workgroup-per-group controller it was
// VALU- and latency-bound on a single CU: 21-54 us per attempt against 10 + 6 for controller + this kernel).
template <int UNROLL, int PER>  // PER outputs per thread: gridDim.z = 3 / PER
__global__ __launch_bounds__(256) void rk45_embed_kernel(const Rk45State *st, int lo, gp_scorenet net, float *tvec) {
    __shared__ float four[128], tf[128];
    st += blockIdx.y;
    if (lo > 0 && st->status != 0) return;  // finished (or failed): no further stage runs for this group
    const int tid = threadIdx.x, slot = lo + blockIdx.x;
    const float tv = st->stage_t[slot];
    if (tid < 64) {
        const float xp = ((tv * net.fourier_w[tid]) * 2.0f) * 3.14159274101257324f;  // f32 evaluation order of the reference
        four[tid] = sinf(xp);
        four[tid + 64] = cosf(xp);
    }
    __syncthreads();
    if (tid < 128) {
        float acc = 0.f;
        const float *w = net.w_t1 + tid;
#pragma unroll UNROLL
        for (int k = 0; k < 128; ++k) acc = fmaf(four[k], w[k * 128], acc);
        tf[tid] = fmaxf(acc + net.b_t1[tid], 0.f);
    }
    __syncthreads();
    float acc[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) acc[u] = 0.f;
    const float *w = net.w_headt + 256 * PER * blockIdx.z + tid;
#pragma unroll UNROLL
    for (int k = 0; k < 128; ++k)
#pragma unroll
        for (int u = 0; u < PER; ++u) acc[u] = fmaf(tf[k], w[k * HEADS + 256 * u], acc[u]);
#pragma unroll
    for (int u = 0; u < PER; ++u) tvec[((size_t)blockIdx.y * 8 + slot) * HEADS + 256 * (PER * blockIdx.z + u) + tid] = acc[u];
}

// Prepare the next attempt: clamp the step to t_bound (rk.py:119-142) - thread 0 ...
__device__ void begin_attempt(Rk45State *st) {
    const double t = st->t, dir = st->direction;
    const double min_step = 10.0 * fabs(nextafter(t, dir * INFINITY) - t);
    double h_abs = st->h_abs;
    if (h_abs < min_step) {
        if (st->step_rejected) {  // inside the rejection loop scipy fails here (rk.py:132-133)
            st->status = -1;
            return;
        }
        h_abs = min_step;  // start of _step_impl: clamp (rk.py:123-124)
    }
    double h = h_abs * dir, t_new = t + h;
    if (dir * (t_new - st->t_bound) > 0) t_new = st->t_bound;
    h = t_new - t;
    st->t_new = t_new;
    st->h = h;
    st->h_abs = fabs(h);
}
// ... and the six stage times, one thread each (f64 pow / log per slot)
__device__ void publish_attempt(Rk45State *st) {
    __syncthreads();  // thread 0's controller state is visible
    if (st->status != 0) return;  // finished or failed (uniform): no further stage runs
    if (threadIdx.x < 6) set_stage(st, threadIdx.x + 1, st->t + DP_C[threadIdx.x + 1] * st->h);
}

// mode 0: after f0 (d0, d1 -> h0, stage slot 0 = t0 + h0*dir);  mode 1: after f1 (d2 -> h_abs, first attempt);
// mode 2: after an attempt (error norm -> accept / reject -> next attempt)
__global__ __launch_bounds__(256) void rk45_decide_kernel(OdeArgs a, int mode) {
    __shared__ double sh[8];
    Rk45State *st = a.st + blockIdx.x;  // one workgroup per group
    // (head-split plan: hsplit partial sums per tile, consecutive - a group's partials stay one contiguous range)
    const int blk0 = (a.grp_info ? a.grp_info[4 * blockIdx.x] : blockIdx.x * a.bpg) * a.hsplit;
    const int nblk = (a.grp_info ? a.grp_info[4 * blockIdx.x + 1] : a.bpg) * a.hsplit;
    const int grows = a.grp_info ? a.grp_info[4 * blockIdx.x + 2] : a.rows_per_group;
    const double *part = a.partials + blk0;
    if (grows <= 0) {  // padding group of a ragged launch: nothing to integrate
        if (threadIdx.x == 0) st->status = 1;
        return;
    }
    const double nn = (double)(a.ext_sums ? a.ext_rows : grows) * (double)a.ncomp;  // size of the state vector the RMS norms run over
    const int ng = gridDim.x;
    if (mode == 0) {
        const double s0 = a.ext_sums ? a.ext_sums[blockIdx.x] : sum_partials(part, nblk, sh);
        const double s1 = a.ext_sums ? a.ext_sums[ng + blockIdx.x] : sum_partials(part + a.nblocks * a.hsplit, nblk, sh);
        if (threadIdx.x == 0) {
            const double d0 = sqrt(s0) / sqrt(nn), d1 = sqrt(s1) / sqrt(nn);  // norm(x) = |x|_2 / sqrt(size)
            const double interval = fabs(st->t_bound - st->t);
            double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
            h0 = fmin(h0, interval);
            st->d0 = d0, st->d1 = d1, st->h0 = h0;
            set_stage(st, 0, st->t + h0 * st->direction);
            st->nfev = 1;
        }
    } else if (mode == 1) {
        const double s0 = a.ext_sums ? a.ext_sums[blockIdx.x] : sum_partials(part, nblk, sh);
        if (threadIdx.x == 0) {
            const double d2 = (sqrt(s0) / sqrt(nn)) / st->h0;
            const double d1 = st->d1;
            double h1;
            if (d1 <= 1e-15 && d2 <= 1e-15)
                h1 = fmax(1e-6, st->h0 * 1e-3);
            else
                h1 = pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
            const double interval = fabs(st->t_bound - st->t);
            st->h_abs = fmin(fmin(100.0 * st->h0, h1), interval);
            st->nfev = 2;
            st->last_accepted = 0;
            st->step_rejected = 0;
            begin_attempt(st);
        }
        publish_attempt(st);
    } else {
        if (st->status != 0) return;
        if (st->xfail) {  // shared-chunk plan: a stage never saw its predecessor (bounded wait) - fail the solve, do not integrate garbage
            if (threadIdx.x == 0) st->status = -2;
            return;
        }
        const double s0 = a.ext_sums ? a.ext_sums[blockIdx.x] : sum_partials(part, nblk, sh);
        if (threadIdx.x == 0) {
            const double err = sqrt(s0) / sqrt(nn);
            const int ia = st->n_attempts;
            if (ia < 512) {
                st->log_t[ia] = st->t;
                st->log_h[ia] = st->h;
                st->log_err[ia] = err;
                st->log_acc[ia] = err < 1.0;
            }
            st->n_attempts = ia + 1;
            st->nfev += 6;
            st->err_norm = err;
            double h_abs = st->h_abs;
            if (err < 1.0) {
                double factor = (err == 0.0) ? MAX_FACTOR : fmin(MAX_FACTOR, SAFETY * pow(err, -0.2));
                if (st->step_rejected) factor = fmin(1.0, factor);
                h_abs *= factor;
                st->t_old = st->t;
                st->h_acc = st->h;
                st->t = st->t_new;  // rk.py:169 (clamped to t_bound in begin_attempt; t + h can miss it by an ulp)
                if (st->n_eval > 0) {   // ivp.py: every not-yet-emitted t_eval point the step has passed (inclusive of t_new)
                    int m = st->next_eval;
                    st->emit_begin = m;
                    while (m < st->n_eval && st->direction * (st->t_eval[m] - st->t) <= 0) ++m;
                    st->emit_end = m;
                    st->next_eval = m;
                }
                st->last_accepted = 1;
                st->step_rejected = 0;
                st->n_accepted += 1;
                if (st->direction * (st->t - st->t_bound) >= 0) st->status = 1;
            } else {
                h_abs *= fmax(MIN_FACTOR, SAFETY * pow(err, -0.2));
                st->last_accepted = 0;
                st->step_rejected = 1;
            }
            st->h_abs = h_abs;
            if (st->status == 0) begin_attempt(st);
        }
        publish_attempt(st);
    }
}

// Sharded batch: this rank's per-group sums of the stage kernels' partials (fixed order) -> ext_sums, for the caller's all-reduce.
__global__ __launch_bounds__(256) void rk45_group_sums_kernel(OdeArgs a, int nsums) {
    __shared__ double sh[8];
    const int blk0 = (a.grp_info ? a.grp_info[4 * blockIdx.x] : blockIdx.x * a.bpg) * a.hsplit;
    const int nblk = (a.grp_info ? a.grp_info[4 * blockIdx.x + 1] : a.bpg) * a.hsplit;
    for (int w = 0; w < nsums; ++w) {
        const double s = sum_partials(a.partials + (size_t)w * a.nblocks * a.hsplit + blk0, nblk, sh);
        if (threadIdx.x == 0) a.ext_sums[(size_t)w * gridDim.x + blockIdx.x] = s;
    }
}

// Record the accepted state (raw) into the trajectory; runs after decide (multi-block, elementwise).
__global__ void rk45_record_kernel(OdeArgs a) {
    const Rk45State *st = a.st + blockIdx.y;  // grid (64, ngroups): every group records its own rows at its own slot
    if (!a.traj || !st->last_accepted || st->status < 0) return;
    const size_t n = (size_t)a.nrows * a.ncomp;
    const size_t g_row0 = a.grp_info ? a.grp_info[4 * blockIdx.y + 3] : (size_t)blockIdx.y * a.rows_per_group;
    const size_t g_rows = a.grp_info ? a.grp_info[4 * blockIdx.y + 2] : a.rows_per_group;
    const size_t e_lo = g_row0 * a.ncomp, e_hi = e_lo + g_rows * a.ncomp;
    if (g_rows == 0) return;
    if (st->n_eval > 0) {
        // 4th-order dense output of the step just accepted (rk.py RkDenseOutput): y(t) = y_old + h * Q . [x, x^2, x^3, x^4],
        // Q = K^T P, x = (t - t_old) / h.  y is still y_old here (the commit happens in the next attempt's first stage).
        const int mb = st->emit_begin, me = st->emit_end;
        if (mb >= me) return;
        const double h = st->h_acc, t_old = st->t_old;
        for (size_t e = e_lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_hi; e += (size_t)gridDim.x * blockDim.x) {
            double Q[4] = {0, 0, 0, 0};
#pragma unroll
            for (int sgi = 0; sgi < 7; ++sgi) {
                const double kv = a.K[(size_t)sgi * n + e];
#pragma unroll
                for (int c = 0; c < 4; ++c) Q[c] += kv * st->Pm[sgi][c];
            }
            const double y0 = a.y[e];
            for (int m = mb; m < me; ++m) {
                const double x = (st->t_eval[m] - t_old) / h;
                const double x2 = x * x;
                a.traj[(size_t)m * n + e] = y0 + h * (Q[0] * x + Q[1] * x2 + Q[2] * (x2 * x) + Q[3] * (x2 * x2));
            }
        }
        return;
    }
    const int slot = st->n_accepted;  // slot 0 holds y0
    if (slot >= st->traj_cap) return;
    for (size_t e = e_lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_hi; e += (size_t)gridDim.x * blockDim.x)
        a.traj[(size_t)slot * n + e] = a.ynew[e];
}

// Denoise (samplers.py:209-218) + normalize_rotation + centre (:224-226); also post-processes the trajectory.
template <int P, int MODEL>
__global__ __launch_bounds__(TrunkCfg<P>::NT) void rk45_finish_kernel(OdeArgs a, gp_scorenet net, double denoise_scale, int do_denoise, double *x_out) {
    using L = TrunkLds<P, OdeModel<MODEL>::BWD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    int grp, row0, rend;
    ode_block<P>(a, blockIdx.x, grp, row0, rend);
    if (row0 >= rend) return;
    const Rk45State *st = a.st + grp;
    const double *yfin = st->last_accepted ? a.ynew : a.y;
    const float *tvec = a.tvec + (size_t)grp * 8 * HEADS;  // slot 0 = eps
    TrunkPre<P> pre;
    trunk_begin<P>(net, pre, a.cvec, tvec, row0, rend, a.kcand);
    const float sigma = st->stage_sigma[0];
    if (tid < P) {
        const int r = row0 + tid < rend ? row0 + tid : rend - 1;
        float *xr = lds + tid * L::LD0;
#pragma unroll
        for (int j = 0; j < 9; ++j) xr[j] = (float)yfin[(size_t)r * 9 + j];  // x.float() (:214)
#pragma unroll
        for (int j = 9; j < 16; ++j) xr[j] = 0.f;
    }
    __syncthreads();
    const float *F;
    int ldf;
    if constexpr (MODEL == 0) {
        trunk_ftheta<P>(lds, net, a.cvec, tvec, row0, rend, a.kcand, pre);
        F = lds + L::OFF_H1, ldf = L::LDH;
    } else {
        F = gp_bwd::score_vjp_tile<gp_bwd::ENERGY>(lds, net, a.cvec, tvec, row0, rend, a.kcand, pre, sigma);
        ldf = gp_bwd::LDS_OUT;
    }
    if (tid < P && row0 + tid < rend) {
        const int r = row0 + tid;
        double xv[9];
        // vec_eps is an f32 [R,1] tensor: g = sigma * sqrt(2 ln 5000) evaluated in f32 (sde.py:20-24)
        const float g = sigma * 4.1272735595703125f;  // (float)sqrt(2*(ln 50 - ln 0.01))
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float grad = MODEL == 0 ? F[tid * ldf + j] / (sigma + 1e-7f) : F[tid * ldf + j];
            const float drift = 0.f - (g * g) * grad;                    // R-SDE sign as written (:216)
            const float dx = drift * (float)denoise_scale;               // f32 tensor * python float stays f32
            xv[j] = yfin[(size_t)r * 9 + j] + (do_denoise ? (double)dx : 0.0);
        }
        normalize_rot6<double>(xv);
        const float *cen = a.centre + (size_t)(r / a.kcand) * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) xv[6 + j] += (double)cen[j];
#pragma unroll
        for (int j = 0; j < 9; ++j) x_out[(size_t)r * 9 + j] = xv[j];
    }
}

// likelihood ODE: the final state [R][10] (pose at t = 1 and the accumulated log-density change) as it is
__global__ void rk45_copy_final_kernel(OdeArgs a, double *x_out) {
    const size_t g_row0 = (size_t)blockIdx.y * a.rows_per_group;
    const Rk45State *st = a.st + blockIdx.y;
    const double *yfin = st->last_accepted ? a.ynew : a.y;
    const size_t e_lo = g_row0 * a.ncomp, e_hi = e_lo + (size_t)a.rows_per_group * a.ncomp;
    for (size_t e = e_lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_hi; e += (size_t)gridDim.x * blockDim.x) x_out[e] = yfin[e];
}

// normalize_rotation + centre on every recorded state (samplers.py:220-224)
__global__ void rk45_traj_post_kernel(int nrows, int kcand, int nstates, const float *__restrict__ centre, double *__restrict__ traj) {
    const size_t total = (size_t)nstates * nrows;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i % nrows);
        double *v = traj + i * 9;
        double xv[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) xv[j] = v[j];
        normalize_rot6<double>(xv);
        const float *cen = centre + (size_t)(r / kcand) * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) xv[6 + j] += (double)cen[j];
#pragma unroll
        for (int j = 0; j < 9; ++j) v[j] = xv[j];
    }
}

__device__ void reset_state(Rk45State *st, double t0, double t_bound, double rtol, double atol, int traj_cap) {
    st->t = t0, st->t_bound = t_bound, st->direction = t_bound >= t0 ? 1.0 : -1.0;
    st->rtol = rtol, st->atol = atol;
    st->h = 0, st->h_abs = 0, st->d0 = st->d1 = st->h0 = 0, st->err_norm = 0;
    st->status = 0, st->last_accepted = 0, st->step_rejected = 0;
    st->n_attempts = 0, st->n_accepted = 0, st->nfev = 0, st->traj_cap = traj_cap;
    st->n_eval = 0, st->next_eval = 0, st->emit_begin = 0, st->emit_end = 0, st->t_old = t0, st->t_eval = nullptr;
    for (int i = 0; i < 64; ++i) st->xflags[i] = 0;
    st->xfail = 0;
    // the very first evaluation gets a python-float t: sde_coeff(torch.tensor(t)) is f32 there;
    // the f64 formula differs by <= 1e-7 relative - documented deviation.
    for (int i = 0; i < 8; ++i) st->stage_t[i] = 0.f, st->stage_sigma[i] = 1.f, st->stage_g2[i] = 0.0;
    set_stage(st, 0, t0);
}
__global__ void rk45_reset_kernel(Rk45State *st, double t0, double t_bound, double rtol, double atol, int traj_cap) {
    if (threadIdx.x == 0) reset_state(st + blockIdx.x, t0, t_bound, rtol, atol, traj_cap);  // one block per group
}

struct DenseP {
    double p[7][4];
};
__global__ void rk45_set_dense_kernel(Rk45State *st, const double *t_eval, int n_eval, DenseP P) {
    if (threadIdx.x != 0) return;
    st += blockIdx.x;
    st->t_eval = t_eval, st->n_eval = n_eval, st->next_eval = 0, st->emit_begin = st->emit_end = 0;
    for (int i = 0; i < 7; ++i)
        for (int c = 0; c < 4; ++c) st->Pm[i][c] = P.p[i][c];
}

__global__ void rk45_set_slot0_kernel(Rk45State *st, double t) {
    if (threadIdx.x == 0) set_stage(st + blockIdx.x, 0, t);
}

template <typename K>
int set_lds_attr(K kern, size_t lds) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? GP_OK
                                                                                                                                            : GP_ELAUNCH;
}

}  // namespace

// CHAIN: the stage kernels run in the chain form (a.bpg / a.nblocks count 128-row workgroups); the denoising evaluation of phase 5
// stays on P-row tiles with its own workgroup count
// SPLIT: the head-split plan (three workgroups per 16-row tile, one head each; OdeArgs::hsplit == 3): every stage is a launch of its own
template <int P, int MODEL, bool CHAIN = false, bool SPLIT = false>
static int rk45_phase_impl(int phase, OdeArgs &a, const gp_scorenet *net, double *traj, int traj_cap, double t0, double t_bound, double rtol,
                           double atol, double denoise_scale, int do_denoise, int nstates, const float *centre, double *x_out, hipStream_t st) {
    const size_t chain_lds = MODEL == 0 ? gp_chain::Cfg<2>::LDS_BYTES : gp_chain::CfgV<2>::LDS_BYTES;
    const double *y = a.y;
    const size_t lds = MODEL == 0 ? trunk_lds_bytes<P>() : gp_bwd::LDS_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        if constexpr (CHAIN) {
            if (set_lds_attr(rk45_stage_chain_kernel<2, 0, MODEL>, chain_lds) || set_lds_attr(rk45_stage_chain_kernel<2, 1, MODEL>, chain_lds) ||
                set_lds_attr(rk45_stage_chain_kernel<2, 2, MODEL>, chain_lds) || set_lds_attr(rk45_stage_chain_kernel<2, 3, MODEL>, chain_lds) ||
                set_lds_attr(rk45_stage_chain_kernel<2, 4, MODEL>, chain_lds) || set_lds_attr(rk45_stage_chain_kernel<2, 5, MODEL>, chain_lds) ||
                set_lds_attr(rk45_stage_chain_kernel<2, 6, MODEL>, chain_lds) || set_lds_attr(rk45_stage_chain_kernel<2, 7, MODEL>, chain_lds))
                return GP_ELAUNCH;
        }
        if (set_lds_attr(rk45_stage_kernel<P, 0, MODEL, SPLIT>, lds) || set_lds_attr(rk45_stage_kernel<P, 1, MODEL, SPLIT>, lds) ||
            set_lds_attr(rk45_stage_kernel<P, 2, MODEL, SPLIT>, lds) || set_lds_attr(rk45_stage_kernel<P, 3, MODEL, SPLIT>, lds) ||
            set_lds_attr(rk45_stage_kernel<P, 4, MODEL, SPLIT>, lds) || set_lds_attr(rk45_stage_kernel<P, 5, MODEL, SPLIT>, lds) ||
            set_lds_attr(rk45_stage_kernel<P, 6, MODEL, SPLIT>, lds) || set_lds_attr(rk45_stage_kernel<P, 7, MODEL, SPLIT>, lds))
            return GP_ELAUNCH;
        if constexpr (MODEL != 2) {
            if (set_lds_attr(rk45_finish_kernel<P, MODEL>, lds)) return GP_ELAUNCH;
        }
        if constexpr (!CHAIN && !SPLIT && P == 16) {
            if (set_lds_attr(rk45_attempt_kernel<P, MODEL>, lds)) return GP_ELAUNCH;
        }
        attr_done = true;
    }
    if ((a.hsplit == 3) != SPLIT) return GP_EINVAL;
    const dim3 grid(a.nblocks * (SPLIT ? 3 : 1)), blk(TrunkCfg<P>::NT), blk1(256);
    const size_t n = (size_t)a.nrows * a.ncomp;
    // time embeddings of stage slots [lo, lo + n) of every group, behind the kernel that decided their times
    auto embed = [&](int lo, int n) {
        if (n * a.ngroups * 3 <= 256)
            hipLaunchKernelGGL((rk45_embed_kernel<64, 1>), dim3(n, a.ngroups, 3), blk1, 0, st, a.st, lo, *net, a.tvec);
        else
            hipLaunchKernelGGL((rk45_embed_kernel<4, 3>), dim3(n, a.ngroups, 1), blk1, 0, st, a.st, lo, *net, a.tvec);
    };
    auto stage = [&](auto tag) {
        constexpr int S = decltype(tag)::value;
        if constexpr (CHAIN)
            hipLaunchKernelGGL((rk45_stage_chain_kernel<2, S, MODEL>), grid, dim3(gp_chain::NT), chain_lds, st, a, *net);
        else
            hipLaunchKernelGGL((rk45_stage_kernel<P, S, MODEL, SPLIT>), grid, blk, lds, st, a, *net);
    };
    switch (phase) {
        case 0:
            hipLaunchKernelGGL(rk45_reset_kernel, dim3(a.ngroups), dim3(64), 0, st, a.st, t0, t_bound, rtol, atol, traj_cap);
            embed(0, 1);
            if (traj && hipMemcpyAsync(traj, y, n * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return GP_ELAUNCH;
            break;
        // with ext_sums (a batch sharded over several GPUs) phases 1-3 stop after the per-group sums and phases 11-13 run the
        // controller on the all-reduced sums
        case 1:
            stage(std::integral_constant<int, 0>{});
            if (a.ext_sums)
                hipLaunchKernelGGL(rk45_group_sums_kernel, dim3(a.ngroups), blk1, 0, st, a, 2);
            else
                hipLaunchKernelGGL(rk45_decide_kernel, dim3(a.ngroups), blk1, 0, st, a, 0);
            embed(0, 1);
            break;
        case 11:
            if (!a.ext_sums) return GP_EINVAL;
            hipLaunchKernelGGL(rk45_decide_kernel, dim3(a.ngroups), blk1, 0, st, a, 0);
            embed(0, 1);
            break;
        case 2:
            stage(std::integral_constant<int, 7>{});
            if (a.ext_sums)
                hipLaunchKernelGGL(rk45_group_sums_kernel, dim3(a.ngroups), blk1, 0, st, a, 1);
            else
                hipLaunchKernelGGL(rk45_decide_kernel, dim3(a.ngroups), blk1, 0, st, a, 1);
            embed(1, 6);
            break;
        case 12:
            if (!a.ext_sums) return GP_EINVAL;
            hipLaunchKernelGGL(rk45_decide_kernel, dim3(a.ngroups), blk1, 0, st, a, 1);
            embed(1, 6);
            break;
        case 3:
            if constexpr (!CHAIN && !SPLIT && P == 16) {
                // latency regime: the six stages of the attempt as ONE launch (rk45_attempt_kernel)
                hipLaunchKernelGGL((rk45_attempt_kernel<P, MODEL>), grid, blk, lds, st, a, *net);
            } else {
                stage(std::integral_constant<int, 1>{});
                stage(std::integral_constant<int, 2>{});
                stage(std::integral_constant<int, 3>{});
                stage(std::integral_constant<int, 4>{});
                stage(std::integral_constant<int, 5>{});
                stage(std::integral_constant<int, 6>{});
            }
            if (a.ext_sums) {
                hipLaunchKernelGGL(rk45_group_sums_kernel, dim3(a.ngroups), blk1, 0, st, a, 1);
                break;
            }
            hipLaunchKernelGGL(rk45_decide_kernel, dim3(a.ngroups), blk1, 0, st, a, 2);
            embed(1, 6);
            if (traj) hipLaunchKernelGGL(rk45_record_kernel, dim3(64, a.ngroups), blk1, 0, st, a);
            break;
        case 13:
            if (!a.ext_sums) return GP_EINVAL;
            hipLaunchKernelGGL(rk45_decide_kernel, dim3(a.ngroups), blk1, 0, st, a, 2);
            embed(1, 6);
            if (traj) hipLaunchKernelGGL(rk45_record_kernel, dim3(64, a.ngroups), blk1, 0, st, a);
            break;
        case 4:
            hipLaunchKernelGGL(rk45_set_slot0_kernel, dim3(a.ngroups), dim3(64), 0, st, a.st, t0);
            embed(0, 1);
            break;
        case 5:
            if (!x_out) return GP_EINVAL;
            if constexpr (MODEL == 2) {
                hipLaunchKernelGGL(rk45_copy_final_kernel, dim3(32, a.ngroups), blk1, 0, st, a, x_out);
            } else {
                OdeArgs af = a;  // (the denoising evaluation runs on whole tiles, one workgroup each, under every plan)
                if constexpr (CHAIN) af.bpg = (a.rows_per_group + P - 1) / P, af.nblocks = af.bpg * a.ngroups;
                hipLaunchKernelGGL((rk45_finish_kernel<P, MODEL>), dim3(af.nblocks), blk, lds, st, af, *net, denoise_scale, do_denoise, x_out);
                if (traj && nstates > 0)
                    hipLaunchKernelGGL(rk45_traj_post_kernel, dim3(128), blk1, 0, st, a.nrows, a.kcand, nstates, centre, traj);
            }
            break;
        default:
            return GP_EINVAL;
    }
    return gp_launch_status();
}

// Shared-chunk plan: which launches of `nrows` rows it serves, and its geometry.  T 16-row chunks on C CUs: `whole` = T / C chunks per
// workgroup (1 or 3: the 16- and 48-row tiles run on four waves like the shared 16-row tile), rem = T % C shared chunks, 6 rem <= C units.
struct SharedPlan {
    int pw, nwg, nshared;
};
static inline bool shared_plan(int nrows, int kcand, SharedPlan *sp) {
    const int C = gp_num_cus(), T = (nrows + 15) / 16;
    const int whole = T / C, rem = T % C;
    if ((whole != 1 && whole != 3) || rem == 0 || 6 * rem > C || rem > 64 || kcand < 24) return false;  // (k >= 24: a 48-row tile spans <= 3 clouds)
    if (sp) sp->pw = 16 * whole, sp->nwg = C, sp->nshared = rem;
    return true;
}
// One attempt under it: the attempt kernel, the controller over the C + rem partial sums, the next attempt's time embeddings, the record.
template <int PW>
static int rk45_attempt_shared(const OdeArgs &a0, const SharedPlan &sp, const gp_scorenet *net, double *traj, hipStream_t st) {
    const size_t lds = trunk_lds_bytes<PW>() > trunk_lds_bytes<16>() ? trunk_lds_bytes<PW>() : trunk_lds_bytes<16>();
    static bool attr_done = false;
    if (!attr_done) {
        if (set_lds_attr(rk45_attempt_shared_kernel<PW>, lds)) return GP_ELAUNCH;
        attr_done = true;
    }
    OdeArgs a = a0;
    a.hsplit = 1, a.bpg = 1 << 30, a.nblocks = sp.nwg + sp.nshared;  // one group: every tile belongs to group 0
    hipLaunchKernelGGL((rk45_attempt_shared_kernel<PW>), dim3(sp.nwg), dim3(TrunkCfg<PW>::NT), lds, st, a, *net, sp.nshared);
    OdeArgs ad = a;
    ad.bpg = ad.nblocks;  // the controller sums the group's C + rem partials
    hipLaunchKernelGGL(rk45_decide_kernel, dim3(1), dim3(256), 0, st, ad, 2);
    hipLaunchKernelGGL((rk45_embed_kernel<64, 1>), dim3(6, 1, 3), dim3(256), 0, st, a.st, 1, *net, a.tvec);
    if (traj) hipLaunchKernelGGL(rk45_record_kernel, dim3(64, 1), dim3(256), 0, st, ad);
    return gp_launch_status();
}

extern "C" {

int64_t gp_rk45_state_bytes(void) { return (int64_t)sizeof(Rk45State); }

/* Dense-output mode (solve_ivp(..., t_eval=...)): call after phase 0.  t_eval: device array [n_eval] f64 (monotone in the
 * integration direction, as np.linspace(T0, eps, n)); P: the 7x4 dense-output matrix of RK45 in HOST memory (row-major;
 * scipy.integrate RK45.P).  traj must then hold [n_eval][R*9]: slot m receives the interpolated state at t_eval[m]. */
int gp_rk45_set_dense_grouped(int ngroups, void *state, const double *t_eval_dev, int n_eval, const double *P_host, gp_stream_t s) {
    if (ngroups <= 0 || !state || !t_eval_dev || n_eval <= 0 || !P_host) return GP_EINVAL;
    DenseP P;
    for (int i = 0; i < 7; ++i)
        for (int c = 0; c < 4; ++c) P.p[i][c] = P_host[i * 4 + c];
    hipLaunchKernelGGL(rk45_set_dense_kernel, dim3(ngroups), dim3(64), 0, (hipStream_t)s, (Rk45State *)state, t_eval_dev, n_eval, P);
    return gp_launch_status();
}

int gp_rk45_set_dense(void *state, const double *t_eval_dev, int n_eval, const double *P_host, gp_stream_t s) {
    return gp_rk45_set_dense_grouped(1, state, t_eval_dev, n_eval, P_host, s);
}

/* Field offsets for host-side inspection: fills out[0..15] with byte offsets of
 * t, h_abs, status, n_attempts, n_accepted, nfev, err_norm, log_t, log_h, log_err, log_acc, stage_t, last_accepted */
int gp_rk45_state_layout(int64_t *out, int n) {
    if (!out || n < 13) return GP_EINVAL;
    out[0] = offsetof(Rk45State, t);
    out[1] = offsetof(Rk45State, h_abs);
    out[2] = offsetof(Rk45State, status);
    out[3] = offsetof(Rk45State, n_attempts);
    out[4] = offsetof(Rk45State, n_accepted);
    out[5] = offsetof(Rk45State, nfev);
    out[6] = offsetof(Rk45State, err_norm);
    out[7] = offsetof(Rk45State, log_t);
    out[8] = offsetof(Rk45State, log_h);
    out[9] = offsetof(Rk45State, log_err);
    out[10] = offsetof(Rk45State, log_acc);
    out[11] = offsetof(Rk45State, stage_t);
    out[12] = offsetof(Rk45State, last_accepted);
    return GP_OK;
}

static int ode_args(OdeArgs *a, int *tile, int plan, int model, const float *probe, int ngroups, int nclouds_per_group, int k, const float *cvec, float *tvec,
                    const float *centre, void *state, double *y, double *ynew, double *K, double *partials, double *traj, float *x32) {
    if (ngroups <= 0 || nclouds_per_group <= 0 || k <= 0 || !cvec || !tvec || !centre || !state || !y || !ynew || !K || !partials) return GP_EINVAL;
    if (model < 0 || model > 2 || (model == 2 && !probe)) return GP_EINVAL;
    const int rg = nclouds_per_group * k;
    // launch plan of the stage kernels (score_trunk.h: score_plan_rows): 16 / 32-row tiles or the 128-row chain form; plan != 0 forces one
    // plan = 0: a WHOLE-tile plan (one partial sum per tile).  The head-split and shared-chunk plans keep more partial sums and are taken
    // only when the caller asks for them by name - gp_rk45_plan_rows() recommends, gp_rk45_partials_count() sizes the buffer.
    if (plan == 0) plan = model == 0 ? score_plan_rows(ngroups * rg, ngroups > 1 ? rg : 0, k) : score_plan_rows_vjp(ngroups * rg, ngroups > 1 ? rg : 0, k);
    if (plan < 0) return GP_EINVAL;
    const int P = plan & ~GP_PLAN_HEADSPLIT;
    a->hsplit = (plan & GP_PLAN_HEADSPLIT) ? 3 : 1;
    if (a->hsplit == 3 && (model != 0 || P != 16)) return GP_EINVAL;  // one head per workgroup: score model, 16-row tiles
    if (P != 16 && P != 32 && P != 64 && P != 128) return GP_EINVAL;
    if (model != 0 && (P == 32 || P == 64)) return GP_EINVAL;  // forward + backward: 16-row tiles (score_bwd.h) or the 128-row chain form (trunk_chain_vjp.h)
    if (P == 128 && !gp_chain::Cfg<2>::fits(k)) return GP_EINVAL;
    if (ngroups > 1 && rg % P != 0) return GP_EINVAL;  // workgroups must not straddle groups
    a->nrows = ngroups * rg, a->kcand = k;
    a->ngroups = ngroups, a->rows_per_group = rg, a->bpg = (rg + P - 1) / P, a->nblocks = a->bpg * ngroups;
    a->blk_info = nullptr, a->grp_info = nullptr;
    a->cvec = cvec, a->tvec = tvec, a->centre = centre, a->st = (Rk45State *)state;
    a->y = y, a->ynew = ynew, a->K = K, a->partials = partials, a->traj = traj, a->x32 = x32;
    a->probe = probe, a->ncomp = model == 2 ? 10 : 9;
    a->ext_sums = nullptr, a->ext_rows = 0;
    *tile = P;
    return GP_OK;
}

/* Phase driver.  Every phase is a fixed launch sequence on stream s (graph-capturable):
 *   phase 0: reset state (t0 -> t_bound), y must hold y0; copies y0 into traj slot 0 when traj != NULL
 *   phase 1: f0 + d0/d1 -> h0
 *   phase 2: f1 + d2 -> h_abs, first attempt's stage times
 *   phase 3: one attempt: 6 stage kernels + decide + record
 *   phase 4: set slot 0 to `eps_t` (denoise evaluation time)
 *   phase 5: finish: denoise + normalise + centre -> x_out [R,9] f64; post-process nstates trajectory states
 * tvec [ngroups][8][768] is scratch owned by the solver: every kernel that decides stage times (reset, the step controller, phase 4)
 * also writes their time embeddings there (the arithmetic of gp_time_embed), so no launch separates the controller from the next stage.
 * ngroups independent batches (nclouds_per_group clouds each, rows / clouds / state laid out group-major) advance with their OWN
 * step controllers - error norm, accept / reject, step size per group, exactly as separate solve_ivp calls - and share every
 * launch; a finished group's workgroups exit at once.  state: ngroups * gp_rk45_state_bytes(); tvec [ngroups][8][768];
 * partials [3][nblocks], nblocks = ngroups * ceil(rows_per_group / rows per workgroup).  plan (include/genpose_hip.h): rows per workgroup
 * of the stage kernels - 16 / 32 / 64 = tile form (models 1 and 2: 16 only), 128 = the chain form of the trunk (EVERY model since round 4:
 * gp_rk45_plan_rows() also picks it for models 1 and 2 from ~24 600 rows, trunk_chain_vjp.h), 0 = pick a whole-tile plan.  `partials`
 * holds gp_rk45_partials_count(model, plan, ...) doubles; a buffer of 3 x (16-row tiles) doubles is enough for every WHOLE-tile plan
 * (plan = 0 included), NOT for 16 | GP_PLAN_HEADSPLIT (three sums per tile) - which is why plan = 0 never picks that one. */
int gp_rk45_phase_model(int model, int plan, const float *probe, int phase, int ngroups, int nclouds_per_group, int k, const gp_scorenet *net, const float *cvec,
                        float *tvec, const float *centre, void *state, double *y, double *ynew, double *K, double *partials, double *traj,
                        int traj_cap, double t0, double t_bound, double rtol, double atol, double denoise_scale, int do_denoise, int nstates,
                        double *x_out, double *ext_sums, int ext_rows_per_group, gp_stream_t s) {
    OdeArgs a;
    int P = 0;
    if (plan > 0 && (plan & GP_PLAN_SHARED)) {
        // shared-chunk plan: the attempts run on it; the two initial evaluations, the denoising evaluation and the bookkeeping phases
        // run on the whole-tile plan of the same launch size
        SharedPlan sp;
        if (model != 0 || ngroups != 1 || ext_sums || nclouds_per_group <= 0 || k <= 0 || !shared_plan(nclouds_per_group * k, k, &sp) ||
            (plan & ~GP_PLAN_SHARED) != sp.pw)
            return GP_EINVAL;
        const int base = score_plan_rows(nclouds_per_group * k, 0, k);
        if (base < 0) return GP_EINVAL;
        if (phase != 3)
            return gp_rk45_phase_model(model, base, probe, phase, ngroups, nclouds_per_group, k, net, cvec, tvec, centre, state, y, ynew, K, partials, traj, traj_cap,
                                       t0, t_bound, rtol, atol, denoise_scale, do_denoise, nstates, x_out, ext_sums, ext_rows_per_group, s);
        if (ode_args(&a, &P, base, model, probe, ngroups, nclouds_per_group, k, cvec, tvec, centre, state, y, ynew, K, partials, traj, nullptr) != GP_OK || !net)
            return GP_EINVAL;
        return sp.pw == 48 ? rk45_attempt_shared<48>(a, sp, net, traj, (hipStream_t)s) : rk45_attempt_shared<16>(a, sp, net, traj, (hipStream_t)s);
    }
    int rc = ode_args(&a, &P, plan, model, probe, ngroups, nclouds_per_group, k, cvec, tvec, centre, state, y, ynew, K, partials, traj, nullptr);
    if (rc != GP_OK || !net) return GP_EINVAL;
    if (ext_sums && ext_rows_per_group < a.rows_per_group) return GP_EINVAL;
    a.ext_sums = ext_sums, a.ext_rows = ext_rows_per_group;
    if (model != 0 && (!net->w_headx_t || !net->w_pose2_t || !net->w_pose0_t)) return GP_EINVAL;
#define GP_RK45_CALL(PP, MM) \
    rk45_phase_impl<PP, MM>(phase, a, net, traj, traj_cap, t0, t_bound, rtol, atol, denoise_scale, do_denoise, nstates, centre, x_out, (hipStream_t)s)
#define GP_RK45_CHAIN(MM) \
    rk45_phase_impl<16, MM, true>(phase, a, net, traj, traj_cap, t0, t_bound, rtol, atol, denoise_scale, do_denoise, nstates, centre, x_out, (hipStream_t)s)
    if (model == 1) return P == 128 ? GP_RK45_CHAIN(1) : GP_RK45_CALL(16, 1);
    if (model == 2) return P == 128 ? GP_RK45_CHAIN(2) : GP_RK45_CALL(16, 2);
#undef GP_RK45_CHAIN
    if (P == 128)
        return rk45_phase_impl<32, 0, true>(phase, a, net, traj, traj_cap, t0, t_bound, rtol, atol, denoise_scale, do_denoise, nstates, centre, x_out,
                                            (hipStream_t)s);
    if (a.hsplit == 3)
        return rk45_phase_impl<16, 0, false, true>(phase, a, net, traj, traj_cap, t0, t_bound, rtol, atol, denoise_scale, do_denoise, nstates, centre, x_out,
                                                   (hipStream_t)s);
    return P == 16 ? GP_RK45_CALL(16, 0) : (P == 64 ? GP_RK45_CALL(64, 0) : GP_RK45_CALL(32, 0));
#undef GP_RK45_CALL
}

int gp_rk45_plan_rows(int model, int ngroups, int nclouds_per_group, int k) {
    if (ngroups <= 0 || nclouds_per_group <= 0 || k <= 0 || model < 0 || model > 2) return GP_EINVAL;
    const int rg = nclouds_per_group * k;
    if (model != 0) return score_plan_rows_vjp(ngroups * rg, ngroups > 1 ? rg : 0, k);
    SharedPlan sp;
    if (ngroups == 1 && shared_plan(rg, k, &sp)) return sp.pw | GP_PLAN_SHARED;  // a few chunks more than whole rounds of the CUs
    return score_plan_latency(ngroups * rg, ngroups > 1 ? rg : 0, k);
}

/* gp_rk45_plan_rows without the shared-chunk plan (a batch sharded over several GPUs: its controller runs on all-reduced per-group sums) */
int gp_rk45_plan_rows_unshared(int model, int ngroups, int nclouds_per_group, int k) {
    if (ngroups <= 0 || nclouds_per_group <= 0 || k <= 0 || model < 0 || model > 2) return GP_EINVAL;
    const int rg = nclouds_per_group * k;
    if (model != 0) return score_plan_rows_vjp(ngroups * rg, ngroups > 1 ? rg : 0, k);
    return score_plan_latency(ngroups * rg, ngroups > 1 ? rg : 0, k);
}

/* Doubles the `partials` buffer of gp_rk45_phase_model must hold under `plan` (0 = what plan 0 resolves to), every phase included. */
int gp_rk45_partials_count(int model, int plan, int ngroups, int nclouds_per_group, int k) {
    if (ngroups <= 0 || nclouds_per_group <= 0 || k <= 0 || model < 0 || model > 2) return GP_EINVAL;
    const int rg = nclouds_per_group * k;
    auto whole = [&](int p) { return 3 * ngroups * ((rg + p - 1) / p); };
    if (plan == 0) plan = model == 0 ? score_plan_rows(ngroups * rg, ngroups > 1 ? rg : 0, k) : score_plan_rows_vjp(ngroups * rg, ngroups > 1 ? rg : 0, k);
    if (plan < 0) return GP_EINVAL;
    if (plan & GP_PLAN_SHARED) {
        SharedPlan sp;
        if (model != 0 || ngroups != 1 || !shared_plan(rg, k, &sp) || (plan & ~GP_PLAN_SHARED) != sp.pw) return GP_EINVAL;
        const int base = score_plan_rows(rg, 0, k);
        const int a = whole(base), b = 3 * (sp.nwg + sp.nshared);
        return a > b ? a : b;
    }
    const int P = plan & ~GP_PLAN_HEADSPLIT;
    if (P != 16 && P != 32 && P != 64 && P != 128) return GP_EINVAL;
    return whole(P) * ((plan & GP_PLAN_HEADSPLIT) ? 3 : 1);
}

int gp_plan_headsplit_pays(int ntiles16) { return headsplit_pays(ntiles16) ? 1 : 0; }

int gp_rk45_phase_grouped(int phase, int ngroups, int nclouds_per_group, int k, const gp_scorenet *net, const float *cvec, float *tvec,
                          const float *centre, void *state, double *y, double *ynew, double *K, double *partials, double *traj, int traj_cap,
                          double t0, double t_bound, double rtol, double atol, double denoise_scale, int do_denoise, int nstates, double *x_out,
                          gp_stream_t s) {
    // the entry points that predate the plan argument keep the partials size their contract states ([3][ngroups * ceil(rows per group /
    // tile)]): whole tiles - the head-split plan (three partial sums per tile) is reached through gp_rk45_phase_model only
    if (ngroups <= 0 || nclouds_per_group <= 0 || k <= 0) return GP_EINVAL;
    const int rg = nclouds_per_group * k, legacy_plan = score_plan_rows(ngroups * rg, ngroups > 1 ? rg : 0, k);
    if (legacy_plan < 0) return GP_EINVAL;
    return gp_rk45_phase_model(0, legacy_plan, nullptr, phase, ngroups, nclouds_per_group, k, net, cvec, tvec, centre, state, y, ynew, K, partials, traj, traj_cap, t0,
                               t_bound, rtol, atol, denoise_scale, do_denoise, nstates, x_out, nullptr, 0, s);
}

int gp_rk45_phase_ragged(int phase, int ngroups, const int32_t *grp_info, int nblocks, const int32_t *blk_info, int tile, int nclouds_total, int k,
                         const gp_scorenet *net, const float *cvec, float *tvec, const float *centre, void *state, double *y, double *ynew,
                         double *K, double *partials, double *traj, int traj_cap, double t0, double t_bound, double rtol, double atol,
                         double denoise_scale, int do_denoise, int nstates, double *x_out, gp_stream_t s) {
    if (ngroups <= 0 || nblocks <= 0 || !grp_info || !blk_info || (tile != 16 && tile != 32 && tile != (16 | GP_PLAN_HEADSPLIT)) || nclouds_total <= 0 ||
        k <= 0 || !net || !cvec || !tvec || !centre || !state || !y || !ynew || !K || !partials)
        return GP_EINVAL;
    OdeArgs a;
    a.hsplit = (tile & GP_PLAN_HEADSPLIT) ? 3 : 1;
    a.nrows = nclouds_total * k, a.kcand = k, a.nblocks = nblocks;
    a.ngroups = ngroups, a.bpg = 1, a.rows_per_group = 0;
    a.blk_info = blk_info, a.grp_info = grp_info;
    a.cvec = cvec, a.tvec = tvec, a.centre = centre, a.st = (Rk45State *)state;
    a.y = y, a.ynew = ynew, a.K = K, a.partials = partials, a.traj = traj, a.x32 = nullptr;
    a.probe = nullptr, a.ncomp = 9;
    a.ext_sums = nullptr, a.ext_rows = 0;
    if (a.hsplit == 3)
        return rk45_phase_impl<16, 0, false, true>(phase, a, net, traj, traj_cap, t0, t_bound, rtol, atol, denoise_scale, do_denoise, nstates, centre, x_out,
                                                   (hipStream_t)s);
    return tile == 16 ? rk45_phase_impl<16, 0>(phase, a, net, traj, traj_cap, t0, t_bound, rtol, atol, denoise_scale, do_denoise, nstates, centre, x_out,
                                               (hipStream_t)s)
                      : rk45_phase_impl<32, 0>(phase, a, net, traj, traj_cap, t0, t_bound, rtol, atol, denoise_scale, do_denoise, nstates, centre, x_out,
                                               (hipStream_t)s);
}

int gp_rk45_phase(int phase, int nclouds, int k, const gp_scorenet *net, const float *cvec, float *tvec, const float *centre, void *state,
                  double *y, double *ynew, double *K, double *partials, double *traj, int traj_cap, double t0, double t_bound, double rtol,
                  double atol, double denoise_scale, int do_denoise, int nstates, double *x_out, gp_stream_t s) {
    return gp_rk45_phase_grouped(phase, 1, nclouds, k, net, cvec, tvec, centre, state, y, ynew, K, partials, traj, traj_cap, t0, t_bound, rtol, atol,
                                 denoise_scale, do_denoise, nstates, x_out, s);
}

}  // extern "C"
