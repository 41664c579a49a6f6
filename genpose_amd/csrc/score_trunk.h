// Shared trunk of PoseScoreNet / PoseEnergyNet for one tile of pose rows (included by scorenet.hip and rk45.hip).
#pragma once
#include "gp_common.h"

namespace gp_trunk {

constexpr int HID = 256, HEADS = 768, POSE = 9;

// ---------------------------------------------------------------------------------------------- trunk
// Waves per workgroup.  Measured on MI355X (sampler launch, K = 50):
//   16-row tile, R = 3200: 8 waves (two per SIMD) 24.3 us vs 24.3 us with 4 - the tile is bound by the weight stream and its
//                          phases are barrier-locked, a second wave per SIMD has nothing different to overlap with;
//   32-row tile, R = 6400: 8 waves 39.2 us vs 40.9 us with 4 - MFMA-bound, the second wave fills epilogue / barrier bubbles.
//   48-row tile (round 6, the RK45 driver's shared-chunk plan): 4 waves, so that the workgroup can also run a 16-row tile (4 waves) between
//                          its own passes - one wave per SIMD, 12 accumulator fragments per wave.
template <int P>
struct TrunkCfg {
    static constexpr int NW = (P <= 16 || P == 48) ? 4 : 8;
    static constexpr int NV = 16 / NW;   // 16-channel chunks of a 256-wide layer per wave
    static constexpr int NT = 64 * NW;   // threads per workgroup
};

// LDS layout for a P-row tile (floats):
//   X0 [P][16+8] | H1 [P][256+8] | H2 [P][256+8] | red [4*NW][P][12] | wout [9][256] | cvt [2][768+16]        (16-row tiles)
//   X0 [P][16+8] | H1 = H2 [P][256+8]            | red [NW][P][12]   | wout [9][256] | cvt [2][768+16]        (compact form)
// wout / cvt = the head epilogue's operands, staged once per launch while the prologue waits on its own loads:
// cvt[c] = cvec[first cloud of the tile + c] + tvec.  (Read back with broadcast ds_read_b128; fetching them from global
// memory in the epilogue costs 1.1-1.7 k cycles per head, and requesting them under the MFMA loop slows the weight
// stream by more than that - the vector memory path is the loop's bottleneck.)
// Compact form (tiles of >= 32 rows): layer 2 runs IN PLACE (H2 aliases H1; every wave keeps its
// outputs in the accumulators until all waves have read H1) and the four lane groups of a wave are combined in registers
// (v_permlane16/32_swap) before parking: red [NW][P][12].  The 32-row tile takes 64.6 KB instead of 135 KB - a workgroup of another
// kernel (furthest point sampling, an SA chain kernel of the next batch's encoder) can share the CU with it - and the launch got
// 3 % FASTER on its own (77.3 -> 75.0 us at 16 000 rows: 37 KB less LDS traffic per head epilogue).  KEEP (the backward pass of
// gp_score_div reads both hidden activations) keeps H1 and H2 apart.
// clouds whose (cvec + tvec) rows a tile stages in LDS: a tile whose rows span more reads them from global memory in the head epilogue
// (the slow path).  16 / 32 rows: two (k >= 31 candidates per cloud guarantees it for 32 rows); 48 / 64 rows: three (k >= 32).
template <int P>
constexpr int trunk_staged_clouds() { return P >= 48 ? 3 : 2; }

template <int P, bool KEEP = false>
struct TrunkLds {
    static constexpr bool COMPACT = P >= 32, INPLACE = COMPACT && !KEEP;
    static constexpr int LD0 = 16 + GP_LD_PAD, LDH = HID + GP_LD_PAD, LDC = HEADS + 16, NCLD = trunk_staged_clouds<P>();
    static constexpr int OFF_H1 = P * LD0, OFF_H2 = INPLACE ? OFF_H1 : OFF_H1 + P * LDH, OFF_RED = OFF_H2 + P * LDH,
                         NRED = P == 48 ? 8 : (COMPACT ? 1 : 4) * TrunkCfg<P>::NW,  // (48 rows: four waves park the eight partials of ORDER8 below)
                         OFF_WOUT = OFF_RED + NRED * P * 12, OFF_CVT = OFF_WOUT + POSE * HID,
                         TOTAL = OFF_CVT + NCLD * LDC;
};

template <int NV, int NI, int NCLD>
struct TrunkPreT {
    WStages<NV> stA;  // first-layer weights (stages 0,1)
    f32x4 b0[NV];     // first-layer bias fragments
    float bout[NI];   // output bias of the (row, component) entries this thread combines at the end
    // epilogue operands in flight to LDS (trunk_begin requests, trunk_ftheta parks them)
    static constexpr int NWO = (POSE * HID / 4 + 64 * (16 / NV) - 1) / (64 * (16 / NV));   // float4 per thread: w_out
    static constexpr int NCV = (NCLD * HEADS / 4 + 64 * (16 / NV) - 1) / (64 * (16 / NV));  // float4 per thread: cvt
    f32x4 swo[NWO], scv[NCV], stv[NCV];
    int cloud0;      // first cloud of the tile
    bool staged;     // tile spans <= NCLD clouds (else the epilogue reads cvec/tvec from global memory)
};
template <int P>
using TrunkPre = TrunkPreT<TrunkCfg<P>::NV, (P * 9 + TrunkCfg<P>::NT - 1) / TrunkCfg<P>::NT, trunk_staged_clouds<P>()>;

// Entry sequence shared by every kernel that evaluates the trunk: request the first layer's weights and bias
// (call BEFORE any prologue work so the latency overlaps it).
template <int P>
__device__ __forceinline__ void trunk_begin(const gp_scorenet &net, TrunkPre<P> &pre, const float *__restrict__ cvec,
                                            const float *__restrict__ tvec, int row0, int nrows, int kcand) {
    constexpr int NW = TrunkCfg<P>::NW, NV = TrunkCfg<P>::NV, NT = TrunkCfg<P>::NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncl[4] = {wave, wave + NW, wave + 2 * NW, wave + 3 * NW};  // first NV entries are this wave's chunks
    mfma_preload<NV>(pre.stA, net.w_pose0, 1, HID / 16, ncl);
#pragma unroll
    for (int i = 0; i < NV; ++i) pre.b0[i] = *reinterpret_cast<const f32x4 *>(net.b_pose0 + ncl[i] * 16 + 4 * (lane >> 4));
    constexpr int NI = (P * POSE + NT - 1) / NT;
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        pre.bout[q] = net.b_out[(threadIdx.x + q * NT) % POSE];
        gp_pin(pre.bout[q]);
    }
    // epilogue operands -> registers now, LDS later (trunk_ftheta)
    const int rlast = (row0 + P - 1 < nrows ? row0 + P - 1 : nrows - 1);
    pre.cloud0 = row0 / kcand;
    constexpr int NCLD = trunk_staged_clouds<P>();
    pre.staged = rlast / kcand - pre.cloud0 <= NCLD - 1;
#pragma unroll
    for (int q = 0; q < TrunkPre<P>::NWO; ++q) {
        int f = threadIdx.x + q * NT;
        f = f < POSE * HID / 4 ? f : POSE * HID / 4 - 1;
        pre.swo[q] = reinterpret_cast<const f32x4 *>(net.w_out)[f];
    }
    if (pre.staged) {
        const int nclouds = (nrows + kcand - 1) / kcand;
#pragma unroll
        for (int q = 0; q < TrunkPre<P>::NCV; ++q) {
            int f = threadIdx.x + q * NT;
            f = f < NCLD * HEADS / 4 ? f : NCLD * HEADS / 4 - 1;
            const int c = f / (HEADS / 4), o = f - c * (HEADS / 4);
            int cl = pre.cloud0 + c;
            cl = cl < nclouds ? cl : nclouds - 1;
            pre.scv[q] = reinterpret_cast<const f32x4 *>(cvec + (size_t)cl * HEADS)[o];
            pre.stv[q] = reinterpret_cast<const f32x4 *>(tvec)[o];
        }
    }
}

// parks the staged epilogue operands in LDS (visible after the next __syncthreads())
template <int P, bool KEEP = false>
__device__ __forceinline__ void trunk_park_epi(float *lds, TrunkPre<P> &pre) {
    using L = TrunkLds<P, KEEP>;
    constexpr int NT = TrunkCfg<P>::NT;
#pragma unroll
    for (int q = 0; q < TrunkPre<P>::NWO; ++q) {
        const int f = threadIdx.x + q * NT;
        if (f < POSE * HID / 4) reinterpret_cast<f32x4 *>(lds + L::OFF_WOUT)[f] = pre.swo[q];
    }
    if (pre.staged) {
#pragma unroll
        for (int q = 0; q < TrunkPre<P>::NCV; ++q) {
            const int f = threadIdx.x + q * NT;
            if (f < L::NCLD * HEADS / 4) {
                const int c = f / (HEADS / 4), o = f - c * (HEADS / 4);
                *reinterpret_cast<f32x4 *>(lds + L::OFF_CVT + c * L::LDC + 4 * o) = pre.scv[q] + pre.stv[q];
            }
        }
    }
}

// dense 256-wide layer of the trunk on pre-requested weights and bias: out = relu(X W^T + b) -> LDS.
// SYNC_BEFORE_STORE: Ys aliases Xs (in-place layer): every wave has finished reading Xs before any wave overwrites it.
template <int PT, int NW, bool SYNC_BEFORE_STORE = false>
__device__ __forceinline__ void trunk_dense(WStages<16 / NW> &st, const f32x4 (&bias)[16 / NW], const float *Xs, int ld,
                                            const float *__restrict__ Wp, int K, float *Ys, int ldo) {
    constexpr int NV = 16 / NW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc[4] = {wave, wave + NW, wave + 2 * NW, wave + 3 * NW};
    f32x4 acc[4][PT];
    mfma_run<NV, PT>(st, Xs, ld, 0, Wp, gp_round16(K) / 16, HID / 16, nc, acc);
    if constexpr (SYNC_BEFORE_STORE) __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = nc[i] * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            f32x4 v = acc[i][p] + bias[i];
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<f32x4 *>(Ys + (p * 16 + (lane & 15)) * ldo + ch) = v;
        }
    }
}

// sum over the four 16-lane groups of a wave, result in every lane; fixed order ((g0 + g1) + (g2 + g3)): deterministic
__device__ __forceinline__ float lane_groups_sum(float v) {
    const int a = __float_as_int(v);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);  // {g0,g0,g2,g2} , {g1,g1,g3,g3}
    const float s = __int_as_float(r[0]) + __int_as_float(r[1]);
    const int b = __float_as_int(s);
    const auto q = __builtin_amdgcn_permlane32_swap(b, b, false, false);  // {lo,lo} , {hi,hi}
    return __int_as_float(q[0]) + __int_as_float(q[1]);
}

// epilogue operands of one head for this wave's 16-channel chunks: cv = cvec[cloud of the row] + tvec (pre-added)
template <int PT, int NV>
struct HeadOps {
    f32x4 w0[NV], w1[NV], w2[NV], cv[NV][PT];
};

template <int P, int PT, int NV, bool KEEP = false>
__device__ __forceinline__ void head_ops_load(HeadOps<PT, NV> &o, const float *lds, bool staged, const float *__restrict__ cvec,
                                              const float *__restrict__ tvec, int h, const int (&nc)[4], const int (&cloud)[PT], int cloud0) {
    using L = TrunkLds<P, KEEP>;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = nc[i] * 16 + 4 * (lane >> 4);  // 0..767
        const int chh = ch - 256 * h;
        o.w0[i] = *reinterpret_cast<const f32x4 *>(lds + L::OFF_WOUT + (3 * h + 0) * HID + chh);
        o.w1[i] = *reinterpret_cast<const f32x4 *>(lds + L::OFF_WOUT + (3 * h + 1) * HID + chh);
        o.w2[i] = *reinterpret_cast<const f32x4 *>(lds + L::OFF_WOUT + (3 * h + 2) * HID + chh);
        if (staged) {
#pragma unroll
            for (int p = 0; p < PT; ++p) o.cv[i][p] = *reinterpret_cast<const f32x4 *>(lds + L::OFF_CVT + (cloud[p] - cloud0) * L::LDC + ch);
        } else {
            const f32x4 tv = *reinterpret_cast<const f32x4 *>(tvec + ch);
#pragma unroll
            for (int p = 0; p < PT; ++p) o.cv[i][p] = *reinterpret_cast<const f32x4 *>(cvec + (size_t)cloud[p] * HEADS + ch) + tv;
        }
    }
}

struct TrunkNoEmit {};

// f_theta (+ output bias) for the tile's rows -> H1[r*LDH + j], j < 9  (KEEP_H1: -> X0[r*LD0 + 12 + j] instead, so that the
// post-ReLU activations of both hidden layers stay intact in H1 / H2 for a backward pass).
// Preconditions: x rows in X0 (cols 0..8, zero padded to 16) + ONE __syncthreads(); trunk_begin() issued earlier.
// rows >= nrows are clamped duplicates.
// emit(h, chunk index, p-chunk, post-ReLU head activations f32x4, w_out rows of the head for these 4 channels, channel 0..767):
// optional hook on every head-layer fragment (the backward seed of gp_score_div is built there).
// SPLIT (the latency regime's "head-split" plan, GP_PLAN_HEADSPLIT): the workgroup evaluates pose_encoder and ONE head, `hsel` - a tile is
// served by three workgroups on three CUs, each streaming half the weights (0.5 MB instead of 1 MB) and issuing half the MFMAs; only
// components 3 hsel .. 3 hsel + 2 of f_theta are produced (the others are left untouched in H1).  Same MFMA sequence per accumulator, same
// combine order per component: the components a workgroup produces are bit-identical to the unsplit tile's.
// ORDER8 (four-wave tiles: 16 rows on request, 48 rows always): the 256 -> 3 output sums are formed in the ORDER of the eight-wave compact
// tiles (32 / 64 rows) - per head, wave w of eight adds the contributions of chunks w and w + 8, its four lane groups are combined as
// (g0 + g1) + (g2 + g3), the eight wave partials are added in wave order.  A four-wave tile owns chunks w, w + 4, w + 8, w + 12 per wave:
// it keeps TWO running sums (chunks w, w + 8 -> "wave" w; chunks w + 4, w + 12 -> "wave" w + 4), reduces each over the lane groups and
// parks eight partials.  Same products, same additions in the same order: the tile's f_theta equals the 32- / 64-row tiles' BIT FOR BIT,
// which is what lets the RK45 driver's shared-chunk plan (48-row own tiles + 16-row shared tiles) return the whole-tile plans' poses.
template <int P, bool KEEP_H1 = false, class Emit = TrunkNoEmit, bool SPLIT = false, bool ORDER8 = (P == 48)>
__device__ __forceinline__ void trunk_ftheta(float *lds, const gp_scorenet &net, const float *__restrict__ cvec, const float *__restrict__ tvec,
                                             int row0, int nrows, int kcand, TrunkPre<P> &pre, Emit emit = Emit(), int hsel = 0) {
    using L = TrunkLds<P, KEEP_H1>;
    constexpr int PT = P / 16, NW = TrunkCfg<P>::NW, NV = TrunkCfg<P>::NV, NT = TrunkCfg<P>::NT;
    static_assert(!ORDER8 || (NW == 4 && NV == 4 && L::NRED >= 8 && !SPLIT), "ORDER8: a four-wave tile reproducing the eight-wave order");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *X0 = lds, *H1 = lds + L::OFF_H1, *H2 = lds + L::OFF_H2, *red = lds + L::OFF_RED;
    const int ncl[4] = {wave, wave + NW, wave + 2 * NW, wave + 3 * NW};
    int nch[3][4];
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) nch[h][i] = 16 * (SPLIT ? hsel : h) + wave + NW * i;  // head h owns n-chunks [16h, 16h+16); first NV valid
    int cloud[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        int r = row0 + p * 16 + (lane & 15);
        if (r >= nrows) r = nrows - 1;
        cloud[p] = r / kcand;
    }
    trunk_park_epi<P, KEEP_H1>(lds, pre);  // visible after the barrier that follows layer 1
    // ---- layer 1 (9 -> 256); layer 2's first weight stages and bias are requested before it runs
    WStages<NV> stB;
    f32x4 b2[NV];
    mfma_preload<NV>(stB, net.w_pose2, HID / 16, HID / 16, ncl);
#pragma unroll
    for (int i = 0; i < NV; ++i) b2[i] = *reinterpret_cast<const f32x4 *>(net.b_pose2 + ncl[i] * 16 + 4 * (lane >> 4));
    trunk_dense<PT, NW>(pre.stA, pre.b0, X0, L::LD0, net.w_pose0, POSE, H1, L::LDH);
    __syncthreads();
    // ---- layer 2 (256 -> 256); head 0's weights (and, small tile, its epilogue operands) requested before it runs
    WStages<NV> stH[2];
    mfma_preload<NV>(stH[0], net.w_headx, HID / 16, HEADS / 16, nch[0]);
    trunk_dense<PT, NW, L::INPLACE>(stB, b2, H1, L::LDH, net.w_pose2, HID, H2, L::LDH);
    __syncthreads();
    // ---- stacked head layer (256 -> 768) with the 256 -> 3 output layers folded into the epilogue
#pragma unroll
    for (int h = 0; h < (SPLIT ? 1 : 3); ++h) {
        const int hh = SPLIT ? hsel : h;  // the head this iteration evaluates
        f32x4 acc[4][PT];
        // next head's first weight stages are requested before this head runs (hides the cold start)
        if (!SPLIT && h < 2) mfma_preload<NV>(stH[(h + 1) & 1], net.w_headx, HID / 16, HEADS / 16, nch[h + 1]);
        mfma_run<NV, PT>(stH[h & 1], H2, L::LDH, 0, net.w_headx, HID / 16, HEADS / 16, nch[h], acc);
        constexpr int NPS = ORDER8 ? 2 : 1;  // running sums per (p-chunk, component): ORDER8 keeps the even and the odd chunks apart
        float part[NPS][PT][3];  // [sum][p-chunk][component], this wave's n-chunks of head h
#pragma unroll
        for (int e = 0; e < NPS; ++e)
#pragma unroll
            for (int p = 0; p < PT; ++p) part[e][p][0] = part[e][p][1] = part[e][p][2] = 0.f;
        HeadOps<PT, NV> o;
        head_ops_load<P, PT, NV, KEEP_H1>(o, lds, pre.staged, cvec, tvec, hh, nch[h], cloud, pre.cloud0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = ORDER8 ? (i & 1) : 0;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                f32x4 v = acc[i][p] + o.cv[i][p];
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
                if constexpr (!__is_same(Emit, TrunkNoEmit)) emit(hh, i, p, v, o.w0[i], o.w1[i], o.w2[i], nch[h][i] * 16 + 4 * (lane >> 4));
                part[e][p][0] += v.x * o.w0[i].x + v.y * o.w0[i].y + v.z * o.w0[i].z + v.w * o.w0[i].w;
                part[e][p][1] += v.x * o.w1[i].x + v.y * o.w1[i].y + v.z * o.w1[i].z + v.w * o.w1[i].w;
                part[e][p][2] += v.x * o.w2[i].x + v.y * o.w2[i].y + v.z * o.w2[i].z + v.w * o.w2[i].w;
            }
        }
        if constexpr (L::COMPACT || ORDER8) {
            // the 4 channel groups of the wave are summed in registers, one partial per (virtual) wave is parked
#pragma unroll
            for (int e = 0; e < NPS; ++e)
#pragma unroll
                for (int p = 0; p < PT; ++p)
#pragma unroll
                    for (int c = 0; c < 3; ++c) part[e][p][c] = lane_groups_sum(part[e][p][c]);
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < NPS; ++e)
#pragma unroll
                    for (int p = 0; p < PT; ++p)
#pragma unroll
                        for (int c = 0; c < 3; ++c) red[((wave + 4 * e) * P + p * 16 + lane) * 12 + 3 * hh + c] = part[e][p][c];
            }
        } else {
            // every lane parks its partial sums; the 4 channel groups x NW waves are combined below in a fixed order
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int c = 0; c < 3; ++c) red[((wave * 4 + (lane >> 4)) * P + p * 16 + (lane & 15)) * 12 + 3 * hh + c] = part[0][p][c];
        }
    }
    __syncthreads();
    constexpr int NI = (P * POSE + NT - 1) / NT;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int e = tid + it * NT;
        if (e < P * POSE) {
            const int r = e / POSE, j = e - r * POSE;
            if (SPLIT && j / 3 != hsel) continue;  // (uniform control flow is not needed below: the barrier follows the loop)
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < (ORDER8 ? 8 : L::NRED); ++q) v += red[(q * P + r) * 12 + j];
            if constexpr (KEEP_H1)
                X0[r * L::LD0 + 12 + j] = v + pre.bout[it];  // x sits in columns 0..8; 12..20 are padding by now
            else
                H1[r * L::LDH + j] = v + pre.bout[it];  // parked in H1 (free now)
        }
    }
    __syncthreads();
}

template <int P>
__device__ __forceinline__ void load_x_tile(float *lds, const float *__restrict__ x, int row0, int nrows) {
    using L = TrunkLds<P>;
    for (int e = threadIdx.x; e < P * 16; e += TrunkCfg<P>::NT) {
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

// Rows per workgroup of a score launch: which form of the trunk serves `nrows` rows best.
//   16 / 32 / 64  tile form (this file): one tile per workgroup.  With 16 rows every weight fragment feeds one MFMA and the kernel is
//             bound by the CU's L2 -> VGPR streaming rate; with 32 rows it is MFMA-bound; 64 rows (round 4) halve the fixed cost per row
//             again and turn launches of 1-2 partly filled rounds of 32-row tiles (12 800 rows = 1.56 rounds) into one round.
//   128       chain form (trunk_chain.h): 4 waves x 32 rows, activations register-resident, weights through an LDS ring; one
//             workgroup per CU, one pass over the weights per 128 rows.
// Measured on MI355X, K = 50, us per launch (profiles/r4_plans.txt):
//     rows     3200   6400  12800  16000  32000  64000
//     16       26.2   50.4   96.6   95.1  181.1  348
//     32       39.6   39.8   74.9   74.8  146.8  290
//     64       71.2   71.5   72.4   72.1  142.2  283
//     128     136.7  137.4  138.1  137.8  138.5  276
// i.e. a ROUND of one workgroup per CU (256 on MI355X) costs 24.5 / 37.5 / 71.3 / 138.6 us; the plan with the smallest predicted launch time is taken.
constexpr float TILE16_ROUND_US = 24.5f, TILE32_ROUND_US = 37.5f, TILE64_ROUND_US = 71.3f, CHAIN128_ROUND_US = 138.6f;
static inline int score_plan_rows(int nrows, int rows_per_group, int kcand) {
    // a workgroup never straddles two batches; the chain form stages cvec + tvec of <= 4 clouds per workgroup (trunk_chain.h: NCL)
    auto fits = [&](int rows) {
        return (rows_per_group <= 0 || rows_per_group % rows == 0) && (rows < 128 || (rows - 2 + kcand) / kcand + 1 <= 4);
    };
    const float inf = 1e30f;
    const int ncu = gp_num_cus();  // a round = one workgroup per CU
    const int t16 = (nrows + 15) / 16, t32 = (nrows + 31) / 32, t64 = (nrows + 63) / 64, w128 = (nrows + 127) / 128;
    const float c16 = fits(16) ? TILE16_ROUND_US * ((t16 + ncu - 1) / ncu) : inf;
    const float c32 = fits(32) ? TILE32_ROUND_US * ((t32 + ncu - 1) / ncu) : inf;
    const float c64 = fits(64) && kcand >= 32 ? TILE64_ROUND_US * ((t64 + ncu - 1) / ncu) : inf;  // k >= 32: a tile's rows span <= 3 clouds
    const float c128 = fits(128) ? CHAIN128_ROUND_US * ((w128 + ncu - 1) / ncu) : inf;
    int best = 16;
    float cb = c16;
    if (c32 < cb) best = 32, cb = c32;
    if (c64 < cb) best = 64, cb = c64;
    if (c128 < cb) best = 128, cb = c128;
    return cb < inf ? best : -1;
}
// The head-split plan (GP_PLAN_HEADSPLIT, round 5) for the LATENCY regime: a 16-row tile drags the whole 0.53 MFLOP / row network - and
// 1 MB of weights - through ONE CU per evaluation while most of the chip idles (a tracking frame: 16-19 tiles, BASELINE configs[0]: one).
// Three workgroups per tile, each recomputing pose_encoder (25 % of the FLOPs) and owning one head: half the weight stream and half the
// MFMA issue per workgroup.  It pays while every workgroup still gets a CU of its own (tiles x 3 <= CUs); beyond that the recomputation
// (1.5x the work) loses.  Measured: profiles/r5_plans.txt.
static inline bool headsplit_pays(int ntiles16) { return ntiles16 * 3 <= gp_num_cus(); }
// plan for score-model launches whose callers can run head-split (the RK45 driver, the PC step): score_plan_rows, upgraded to
// 16 | GP_PLAN_HEADSPLIT in the latency regime
static inline int score_plan_latency(int nrows, int rows_per_group, int kcand) {
    const int p = score_plan_rows(nrows, rows_per_group, kcand);
    if (p != 16) return p;
    const int tiles = rows_per_group > 0 ? (nrows / rows_per_group) * ((rows_per_group + 15) / 16) : (nrows + 15) / 16;
    return headsplit_pays(tiles) ? (16 | GP_PLAN_HEADSPLIT) : 16;
}
// The same choice for the forward + vector-Jacobian right-hand sides (energy model's score, likelihood ODE): 16-row tiles (score_bwd.h)
// or the 128-row chain form (trunk_chain_vjp.h).  Measured on MI355X, K = 50, us per launch of the energy model's PC step
// (profiles/r4_vjp_plans.txt):
//     rows     3200  12800  32000  64000
//     16       48.1  180.0  372.6  788     (0.45 - 0.58 of the fp32 MFMA peak on 2 x 0.5335 MFLOP per row)
//     128     285.9  287.9  292.2  601     (0.74 at 32 000 rows)
// i.e. a round of one workgroup per CU costs ~47 / ~288 us: the chain form from ~24 600 rows.
constexpr float TILE16_VJP_ROUND_US = 47.0f, CHAIN128_VJP_ROUND_US = 288.0f;
static inline int score_plan_rows_vjp(int nrows, int rows_per_group, int kcand) {
    const bool fits128 = (rows_per_group <= 0 || rows_per_group % 128 == 0) && (128 - 2 + kcand) / kcand + 1 <= 4;
    const bool fits16 = rows_per_group <= 0 || rows_per_group % 16 == 0;
    const int ncu = gp_num_cus();
    const float c16 = fits16 ? TILE16_VJP_ROUND_US * (((nrows + 15) / 16 + ncu - 1) / ncu) : 1e30f;
    const float c128 = fits128 ? CHAIN128_VJP_ROUND_US * (((nrows + 127) / 128 + ncu - 1) / ncu) : 1e30f;
    if (c16 >= 1e30f && c128 >= 1e30f) return -1;
    return c128 < c16 ? 128 : 16;
}
// tile form only (entry points that have no chain form: ragged RK45 groups, the backward kernels)
static inline int score_tile_rows(int nrows) {
    const int ncu = gp_num_cus();
    const int r16 = ((nrows + 15) / 16 + ncu - 1) / ncu, r32 = ((nrows + 31) / 32 + ncu - 1) / ncu;
    return TILE32_ROUND_US * r32 < TILE16_ROUND_US * r16 ? 32 : 16;
}

// One predictor-corrector update of a row (samplers.py:129-152): Langevin corrector with the batch-mean gradient norm `gn`,
// renormalisation of the two rotation columns, Euler-Maruyama predictor with the PRE-corrector score and the reference's sign,
// normalize_rotation.  xv: state in / out; mx: the predictor mean (before its noise; mean_x of the last step).
__device__ __forceinline__ void pc_update_row(float (&xv)[9], const float (&gr)[9], const float (&zz1)[9], const float (&zz2)[9], float gn, float g,
                                              float dt, float sqdt, float (&mx)[9]);

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

__device__ __forceinline__ void pc_update_row(float (&xv)[9], const float (&gr)[9], const float (&zz1)[9], const float (&zz2)[9], float gn, float g,
                                              float dt, float sqdt, float (&mx)[9]) {
    const float q = 0.48f / gn;  // snr * sqrt(pose_dim) = 0.16 * 3
    const float lstep = 2.0f * (q * q);
    const float ns = sqrtf(2.0f * lstep);
#pragma unroll
    for (int j = 0; j < 9; ++j) xv[j] = (xv[j] + lstep * gr[j]) + ns * zz1[j];
    const float n1 = sqrtf(xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2]);
    const float n2 = sqrtf(xv[3] * xv[3] + xv[4] * xv[4] + xv[5] * xv[5]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        xv[j] /= n1;
        xv[3 + j] /= n2;
    }
    const float g2 = g * g;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const float drift = 0.f - g2 * gr[j];  // sign as written in the reference (samplers.py:147)
        mx[j] = xv[j] + drift * dt;
        xv[j] = mx[j] + (g * sqdt) * zz2[j];
    }
    normalize_rot6(xv);
}

}  // namespace gp_trunk
