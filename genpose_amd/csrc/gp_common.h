// Shared device helpers for libgenpose_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/genpose_hip.h"

static inline int gp_launch_status() { return hipGetLastError() == hipSuccess ? GP_OK : GP_ELAUNCH; }

// Compute units of the CURRENT device (MI355X: 256 in 8 XCDs), queried once per device ordinal: persistent grids are sized from it.
// (Per device, not per process: a process may hold partitioned and unpartitioned devices side by side.  The table is filled with plain
// stores of an idempotent value - two threads racing on one slot write the same number.)
static inline int gp_num_cus() {
    constexpr int MAXDEV = 64;
    static int cus[MAXDEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) dev = 0;
    int n = cus[dev];
    if (n == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = n = v;
    }
    return n;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Keeps an early-requested value where it was requested: without it hipcc sinks the load next to its first use.
__device__ __forceinline__ void gp_pin(float &v) { asm volatile("" : "+v"(v)); }

// ------------------------------------------------------------------------------------------------
// fp32 MFMA building block: Y^T[N, P] = W[N, K] . X^T[K, P]  on v_mfma_f32_16x16x4_f32
// (exact f32: bitwise a k-ordered fmaf chain, 157 TFLOP/s chip peak - MI355X_MICROARCH.md).
//
//  A operand = weights.  Packed on the host (gp_pack_weight) so that ONE coalesced global_load_dwordx4
//  per lane feeds four consecutive MFMAs:
//      Wp[((kg*NC + nc)*64 + lane)*4 + jj] = W[nc*16 + (lane&15)][kg*16 + 4*(lane>>4) + jj]
//  (k-group major: the 16-channel chunks of one k-group are CONTIGUOUS, so the waves of a workgroup - and the ~200
//  workgroups that stream the same weights in near lock-step - sweep contiguous memory and spread over all L2 channels;
//  measured equal to chunk-major on MI355X, kept for the contiguous sweep)
//  B operand = activations in LDS, row-major [point][k] with row stride ld = Kpad + 8 floats
//  (ld = 8*odd -> the ds_read_b128 of 16 rows x 4 k-quads is bank-conflict free):
//      lane reads X[p0 + (lane&15)][kg*16 + 4*(lane>>4) .. +3]
//  MFMA #jj of k-group kg therefore contracts k = kg*16 + 4*g + jj (g = 0..3) on both operands.
//  D: lane holds Y[point p0 + (lane&15)][channel n0 + 4*(lane>>4) + r], r = 0..3  -> four consecutive
//  channels of one point: the next layer's operand is written back with one ds_write_b128.
// ------------------------------------------------------------------------------------------------
#define GP_LD_PAD 8

__host__ __device__ static inline int gp_round16(int v) { return (v + 15) & ~15; }

// One wave: NV (<= 4) n-chunks x PT p-chunks of 16x16 outputs, full K loop.
//   Xs: LDS activations, ld floats per row; pc0: first p-chunk of this wave
//   Wp: packed weights; KG k-groups; NC chunks per k-group; nc[i]: n-chunk indices (wave-uniform), i < NV
// Software pipeline: three register stages, fragments for k-group kg+2 are requested while kg is multiplied
// (L2 latency ~ 500-800 cycles vs 128*NV*PT MFMA cycles per k-group).  The stage index is static (unroll by 3), so no
// register rotation copies are needed, and sched_barrier keeps the requests ABOVE the MFMA block (hipcc otherwise
// sinks loads next to their first use and the counted s_waitcnt degenerates to vmcnt(0)).
// mfma_preload() may be issued EARLY (before the producing layer's epilogue / barrier) to hide the cold start.
constexpr int MST = 3;  // register stages of the weight/activation pipeline (prefetch distance = stages - 1 k-groups); 4 and 5 measured slower

template <int NV>
struct WStages {
    f32x4 w[MST][NV];
};

template <int NV>
__device__ __forceinline__ void mfma_preload(WStages<NV> &st, const float *__restrict__ Wp, int KG, int NC, const int (&nc)[4]) {
    const int lane = threadIdx.x & 63;
    const size_t kstride = (size_t)NC * 64;
#pragma unroll
    for (int d = 0; d < MST - 1; ++d) {
        const int k0 = d < KG ? d : KG - 1;
#pragma unroll
        for (int i = 0; i < NV; ++i) st.w[d][i] = (reinterpret_cast<const f32x4 *>(Wp) + (size_t)nc[i] * 64 + lane)[(size_t)k0 * kstride];
    }
}

struct MfmaNoMid {
    __device__ __forceinline__ void operator()() const {}
};

// One pipeline step: request stage (d + MST - 1) % MST for k-group kgd + MST - 1 (clamped), multiply stage d.
template <int NV, int PT, int D>
__device__ __forceinline__ void mfma_step(WStages<NV> &st, f32x4 (&xq)[MST][PT], const f32x4 *const (&wp)[NV], const float *const (&xrow)[PT],
                                          size_t kstride, int kgd, int KG, f32x4 (&acc)[4][PT]) {
    const int nxt = (kgd + MST - 1 < KG) ? kgd + MST - 1 : KG - 1;
    constexpr int e = (D + MST - 1) % MST;
#pragma unroll
    for (int i = 0; i < NV; ++i) st.w[e][i] = wp[i][(size_t)nxt * kstride];
#pragma unroll
    for (int p = 0; p < PT; ++p) xq[e][p] = *reinterpret_cast<const f32x4 *>(xrow[p] + nxt * 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int p = 0; p < PT; ++p) acc[i][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.w[D][i][jj], xq[D][p][jj], acc[i][p], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}

template <int NV, int PT>
__device__ __forceinline__ void mfma_triple(WStages<NV> &st, f32x4 (&xq)[MST][PT], const f32x4 *const (&wp)[NV], const float *const (&xrow)[PT],
                                            size_t kstride, int kg, int KG, f32x4 (&acc)[4][PT]) {
    static_assert(MST >= 2 && MST <= 5, "stage count");
    mfma_step<NV, PT, 0>(st, xq, wp, xrow, kstride, kg, KG, acc);
    mfma_step<NV, PT, 1>(st, xq, wp, xrow, kstride, kg + 1, KG, acc);
    if constexpr (MST > 2) mfma_step<NV, PT, 2>(st, xq, wp, xrow, kstride, kg + 2, KG, acc);
    if constexpr (MST > 3) mfma_step<NV, PT, 3>(st, xq, wp, xrow, kstride, kg + 3, KG, acc);
    if constexpr (MST > 4) mfma_step<NV, PT, 4>(st, xq, wp, xrow, kstride, kg + 4, KG, acc);
}

// `mid` (optional) is invoked once after the k-groups below `kmid` (rounded down to whole stage rounds): loads issued there
// (an epilogue's operands) travel in the shadow of the remaining MFMAs.  The round that follows it is straight-line code, so
// the compiler's counted s_waitcnt stays exact across the extra requests (a loop header would merge to the conservative count
// and stall on them at once).
template <int NV, int PT, class Mid = MfmaNoMid>
__device__ __forceinline__ void mfma_run(WStages<NV> &st, const float *Xs, int ld, int pc0, const float *__restrict__ Wp, int KG, int NC,
                                         const int (&nc)[4], f32x4 (&acc)[4][PT], Mid mid = Mid(), int kmid = 0) {
    const int lane = threadIdx.x & 63;
    const float *xrow[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) xrow[p] = Xs + ((pc0 + p) * 16 + (lane & 15)) * ld + 4 * (lane >> 4);
    const f32x4 *wp[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) wp[i] = reinterpret_cast<const f32x4 *>(Wp) + (size_t)nc[i] * 64 + lane;
    const size_t kstride = (size_t)NC * 64;  // f32x4 units between consecutive k-groups
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xq[MST][PT];
#pragma unroll
    for (int d = 0; d < MST - 1; ++d) {
        const int k0 = d < KG ? d : KG - 1;
#pragma unroll
        for (int p = 0; p < PT; ++p) xq[d][p] = *reinterpret_cast<const f32x4 *>(xrow[p] + k0 * 16);
    }
    // main loop: whole stage rounds, branch-free body (clamped prefetch index: the last requests re-read the final
    // k-group, harmless) so that the compiler's s_waitcnt counts stay exact: vmcnt(2*NV) = "two stages still in flight"
    int kg = 0;
    if constexpr (!__is_same(Mid, MfmaNoMid)) {
#pragma unroll 1
        for (; kg + MST <= kmid && kg + MST <= KG; kg += MST) mfma_triple<NV, PT>(st, xq, wp, xrow, kstride, kg, KG, acc);
        mid();
        __builtin_amdgcn_sched_barrier(0);
        if (kg + MST <= KG) {
            mfma_triple<NV, PT>(st, xq, wp, xrow, kstride, kg, KG, acc);
            kg += MST;
        }
    }
#pragma unroll 1
    for (; kg + MST <= KG; kg += MST) mfma_triple<NV, PT>(st, xq, wp, xrow, kstride, kg, KG, acc);
    // tail: KG % MST k-groups, already resident in stages 0 .. MST-2
#pragma unroll
    for (int d = 0; d < MST - 1; ++d) {
        if (kg + d < KG) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < NV; ++i)
#pragma unroll
                    for (int p = 0; p < PT; ++p)
                        acc[i][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.w[d][i][jj], xq[d][p][jj], acc[i][p], 0, 0, 0);
        }
    }
}

template <int NV, int PT>
__device__ __forceinline__ void mfma_tile(const float *Xs, int ld, int pc0, const float *__restrict__ Wp, int KG, int NC, const int (&nc)[4],
                                          f32x4 (&acc)[4][PT]) {
    WStages<NV> st;
    mfma_preload<NV>(st, Wp, KG, NC, nc);
    mfma_run<NV, PT>(st, Xs, ld, pc0, Wp, KG, NC, nc, acc);
}

// nv-dispatching wrapper: only the valid n-chunks are computed (no wasted MFMAs on ragged channel counts)
template <int PT>
__device__ __forceinline__ void mfma_tile_n(int nv, const float *Xs, int ld, int pc0, const float *__restrict__ Wp, int KG, int NC,
                                            const int (&nc)[4], f32x4 (&acc)[4][PT]) {
    switch (nv) {
        case 1: mfma_tile<1, PT>(Xs, ld, pc0, Wp, KG, NC, nc, acc); break;
        case 2: mfma_tile<2, PT>(Xs, ld, pc0, Wp, KG, NC, nc, acc); break;
        case 3: mfma_tile<3, PT>(Xs, ld, pc0, Wp, KG, NC, nc, acc); break;
        default: mfma_tile<4, PT>(Xs, ld, pc0, Wp, KG, NC, nc, acc); break;
    }
}

// Wave arrangement for a layer with NC output chunks on a P-point tile: WN waves along channels x (4/WN) along points,
// chosen so that every wave gets >= 4 chunks when possible (narrow layers tile the POINT dimension instead).
__device__ __forceinline__ int pick_wn(int NC, int P, int min_pts) {
    int wn = NC >= 16 ? 4 : (NC >= 8 ? 2 : 1);
    while (wn < 4 && (P * wn / 4 < 16 || P * wn / 4 < min_pts)) wn *= 2;  // each wave needs >= 16 (and >= min_pts) points
    return wn;
}

// Dense layer over a P-point LDS tile: out = relu(X W^T + bias) written back to LDS (row stride ldo).
template <int PT, int WN, bool RELU>
__device__ __forceinline__ void dense_to_lds_w(const float *Xs, int ld, const float *__restrict__ Wp, const float *__restrict__ bias, int K,
                                               int N, float *Ys, int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % WN, wp = wave / WN;
    const int KG = gp_round16(K) / 16, NC = gp_round16(N) / 16;
    const int pc0 = wp * PT;
    for (int ncb = wn; ncb < NC; ncb += WN * 4) {
        int nc[4], nv = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            nc[i] = ncb + i * WN;
            nv += nc[i] < NC;
        }
        f32x4 acc[4][PT];
        mfma_tile_n<PT>(nv, Xs, ld, pc0, Wp, KG, NC, nc, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= nv) break;
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

// P-point tile, wave arrangement picked per layer at run time.
template <int P, bool RELU>
__device__ __forceinline__ void dense_to_lds(const float *Xs, int ld, const float *__restrict__ Wp, const float *__restrict__ bias, int K, int N,
                                             float *Ys, int ldo) {
    const int wn = pick_wn(gp_round16(N) / 16, P, 16);
    if constexpr (P >= 64) {
        if (wn == 1) return dense_to_lds_w<P / 64, 1, RELU>(Xs, ld, Wp, bias, K, N, Ys, ldo);
    }
    if constexpr (P >= 32) {
        if (wn == 2) return dense_to_lds_w<P / 32, 2, RELU>(Xs, ld, Wp, bias, K, N, Ys, ldo);
    }
    return dense_to_lds_w<P / 16, 4, RELU>(Xs, ld, Wp, bias, K, N, Ys, ldo);
}

// Wave-wide float sum on DPP row operations (VALU speed, fixed order -> deterministic); result valid in EVERY lane
// (read back from lane 63 as a wave-uniform scalar).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f32(float v) {
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
    return v + o;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v = dpp_add_f32<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dpp_add_f32<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v = dpp_add_f32<0x141, 0xF>(v);  // row_half_mirror
    v = dpp_add_f32<0x140, 0xF>(v);  // row_mirror: every lane of a row holds the row sum
    v = dpp_add_f32<0x142, 0xA>(v);  // row_bcast15 -> rows 1,3 (+ previous row's sum; other rows add 0)
    v = dpp_add_f32<0x143, 0xC>(v);  // row_bcast31 -> rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// max over the 16 lanes that share (lane >> 4): the 16 points of one p-chunk.  DPP row operations (VALU, no LDS traffic: the
// ds_bpermute form of __shfl_xor costs an LDS instruction per step, and the SA epilogues take 64 of these per 16 rows); after
// row_mirror every lane of the row holds the row's maximum.  max is order-independent: same bits as any other reduction order.
template <int CTRL>
__device__ __forceinline__ float dpp_max_f32(float v) {
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
    return fmaxf(v, o);
}
__device__ __forceinline__ float row16_max(float v) {
    v = dpp_max_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_max_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_max_f32<0x141>(v);  // row_half_mirror
    v = dpp_max_f32<0x140>(v);  // row_mirror
    return v;
}
// max over the 8 lanes of each half of a 16-lane row (two 8-row neighbourhoods in one p-chunk: nsample = 8)
__device__ __forceinline__ float row8_max(float v) {
    v = dpp_max_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_max_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_max_f32<0x141>(v);  // row_half_mirror: lane i <-> 7 - i inside each half
    return v;
}
// Max over the 16 points of a TRANSPOSED accumulator fragment.  Calling the MFMA with the operands swapped - activations as A,
// weights as B (the per-lane fragments are the same registers either way) - gives D^T: lane = channel 16 n + (lane & 15), the four
// registers x four lane groups = the 16 points.  The max over the points is then 3 in-lane max + two permlane swaps (7 instructions
// per 16 channels) instead of 4 DPP steps for each of the 4 registers (32); every lane ends up with the maximum.
//
// On this part an fp32 MFMA executes on the SIMD's FMA lanes - every other VALU instruction a wave issues takes four cycles away from the
// matrix work (measured over all kernels of the encoder: time per MFMA = 32 cycles + 4 x the VALU instructions per MFMA), so the
// instruction count of an epilogue IS its cost.  As fmaxf() the compiler must first quiet possible signalling NaNs of every input that is
// not the result of an arithmetic instruction (MFMA results, permlane results: IEEE mode) - `v_max_f32 x, x, x`, 8 of the 15 VALU
// instructions this function used to be.  v_med3_f32(a, b, +inf) is max(a, b) for non-NaN inputs and is selected as written.
// (NOT inline asm: the hazard recogniser does not look into it, and an MFMA result needs up to 11 wait states before a VALU read.)
// (the +inf goes through an empty asm: the optimiser folds fmed3(a, b, +inf) back into maxnum - and its canonicalisation - otherwise)
__device__ __forceinline__ float max_raw(float a, float b) {
    float inf = __builtin_inff();
    asm("" : "+s"(inf));
    return __builtin_amdgcn_fmed3f(a, b, inf);
}
__device__ __forceinline__ float points16_max_t(const f32x4 &v) {
    float m = max_raw(max_raw(v.x, v.y), max_raw(v.z, v.w));
    const int a = __float_as_int(m);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);  // {g0,g0,g2,g2} , {g1,g1,g3,g3}
    m = max_raw(__int_as_float(r[0]), __int_as_float(r[1]));
    const int b = __float_as_int(m);
    const auto q = __builtin_amdgcn_permlane32_swap(b, b, false, false);  // {lo,lo} , {hi,hi}
    return max_raw(__int_as_float(q[0]), __int_as_float(q[1]));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}
