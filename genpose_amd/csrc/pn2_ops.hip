// PointNet++ grouping operators for gfx950 (MI355X): FPS, ball query, gather, group, three-NN, interpolate.
// Replaces the reference's CUDA extension `pointnet2_cuda`
// (networks/pts_encoder/pointnet2_utils/pointnet2/src/*.cu) behind the C ABI of include/genpose_hip.h.
//
// Design (wave64, LDS-staged):
//  * FPS: ONE WAVE per cloud for n <= 1024 (four waves up to 4096 points), coordinates and running min-distances live in registers,
//    the cloud is mirrored in LDS only for the broadcast read of the last selected point; a point's (tie rank, distance bits) is one
//    64-bit register pair, the lane's best pair a tree of v_max_f64, the wave's two six-step v_max_u32_dpp reductions - no LDS
//    exchange and no barrier in the pick loop of the one-wave form (fps_pass).  The tie rank reproduces the reference's
//    shared-memory tree (sampling_gpu.cu:86-91,143-203): smallest bit-reversed slot wins.
//  * ball query: one wave per centre, 64 candidates per step; compare -> v_mbcnt slot -> LDS staging row keeps index order, one
//    coalesced store per centre (ball_query_kernel).
//  * distances: `dx*dx + dy*dy + dz*dz` (sampling_gpu.cu:133, ball_query_gpu.cu:33, interpolate_gpu.cu:36) under one of three
//    contraction conventions (GP_ARITH_A / _B / _C, include/genpose_hip.h) - a TEMPLATE parameter of every kernel that
//    evaluates it (no run-time cost), spelled with __fmaf_rn/__fmul_rn/__fadd_rn so hipcc cannot re-associate or re-contract it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/genpose_hip.h"
#include "gp_common.h"

namespace {

// a0*b0 + a1*b1 + a2*b2 under contraction convention AR (the reference's text leaves the contraction to nvcc):
//   A  fma(a2,b2, fma(a1,b1, a0*b0))     B  fma(a2,b2, fma(a0,b0, a1*b1))  (LLVM / GCC: the first fadd fuses its left multiply)
//   C  (a0*b0 + a1*b1) + a2*b2           (no contraction)
template <int AR>
__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    static_assert(AR == GP_ARITH_A || AR == GP_ARITH_B || AR == GP_ARITH_C, "unknown arithmetic convention");
    if constexpr (AR == GP_ARITH_A) return __fmaf_rn(a2, b2, __fmaf_rn(a1, b1, __fmul_rn(a0, b0)));
    if constexpr (AR == GP_ARITH_B) return __fmaf_rn(a2, b2, __fmaf_rn(a0, b0, __fmul_rn(a1, b1)));
    return __fadd_rn(__fadd_rn(__fmul_rn(a0, b0), __fmul_rn(a1, b1)), __fmul_rn(a2, b2));
}

template <int AR>
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return dot3<AR>(dx, dx, dy, dy, dz, dz);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// the same on the packed-f32 VALU, two points per instruction (v_pk_mul / v_pk_fma / v_pk_add: the IEEE operations of dot3<AR>, in
// its order - identical bits; the library is built with -ffp-contract=off, so convention C's adds stay adds)
template <int AR>
__device__ __forceinline__ f32x2 sqnorm_pk(f32x2 dx, f32x2 dy, f32x2 dz) {
    if constexpr (AR == GP_ARITH_A) return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
    if constexpr (AR == GP_ARITH_B) return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
    const f32x2 s = dx * dx + dy * dy;
    return s + dz * dz;
}

// order-preserving float -> uint (total order on non-NaN floats)
__device__ __forceinline__ uint32_t fkey(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Wave-wide max of a u32 with DPP row operations (VALU speed; no LDS crossbar traffic): quad swaps, row mirrors, then
// row_bcast15 / row_bcast31 accumulate into lane 63, which is read back as a wave-uniform scalar.  Written as six `v_max_u32_dpp`
// (the compiler's form of update_dpp + max is mov / nop / mov_dpp / max: four issue slots per step, and every slot of this chain is
// on the critical path of a pick); a VALU result needs two wait states before a DPP read, hence the s_nop 1 in front of each step.
// All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"   // every lane of a 16-lane row holds the row max
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"  // into rows 1, 3 (other rows keep their value)
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"  // into rows 2, 3 -> lane 63 holds the wave max
        "s_nop 1"
        : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// max of two (distance bits, rank) keys held as hi:lo of a register pair, in ONE instruction: for 0 <= hi < 0x7ff00000 the pair read as a
// double is a non-negative finite number (or a subnormal / zero: fp64 subnormals are never flushed on this target), and for those the order of
// the doubles is the order of the 64-bit patterns - v_max_f64 is the 64-bit unsigned max the ISA lacks.  hi is a running minimum that starts
// at 1e10f (0x501502f9) or at the caller's temp (finite, >= 0), so the exponent field never reaches 0x7ff.  Inline asm: as llvm.maxnum the
// compiler would first canonicalise both operands (IEEE mode), doubling the work.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));  // .x = lo (rank), .y = hi (distance bits): one 64-bit register pair
__device__ __forceinline__ u32x2 max_key(u32x2 a, u32x2 b) {
    u32x2 r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// LDS-qualified pointers: with plain `const float *` the planes travel through a pointer array and the winner's coordinates are
// fetched with flat_load (the slow aperture path, on the critical path of every selected point) instead of ds_read.
typedef __attribute__((address_space(3))) float lds_f32;
typedef const lds_f32 *lds_cptr;
template <typename T>
__device__ __forceinline__ lds_cptr as_lds(const T *p) { return (lds_cptr)(p); }

constexpr int FPS_T = 256;       // threads per cloud of the wide form (n > 1024, and the global-memory fallback)
constexpr int FPS_W = 64;        // ... of the one-wave form (n <= 1024): no cross-wave exchange, no barrier in the pick loop
constexpr int FPS_MAXPPT = 16;   // register-resident points per thread (n <= 1024 on one wave, n <= 4096 on four)

struct FpsRank {
    int S, logS, Q;
    // rank of element k: larger = preferred among equal distances.
    //  in-thread (same k % S): first strict maximum = smaller k wins  (sampling_gpu.cu:124-138)
    //  across threads: lower slot wins at every tree level = smaller bit-reversed slot wins (:86-91,143-203)
    __device__ __forceinline__ uint32_t rank(int k) const {
        uint32_t slot = (uint32_t)k & (uint32_t)(S - 1);
        uint32_t br = logS ? (__brev(slot) >> (32 - logS)) : 0u;
        return (uint32_t)(S - 1 - (int)br) * (uint32_t)Q + (uint32_t)(Q - 1 - (k >> logS));
    }
    __device__ __forceinline__ int unrank(uint32_t r) const {
        if (Q == 1) {  // n <= 1024 and a power of two (every level of the encoder): no division
            const uint32_t br1 = (uint32_t)(S - 1) - r;
            return (int)(logS ? (__brev(br1) >> (32 - logS)) : 0u);
        }
        uint32_t a = r / (uint32_t)Q, bq = r - a * (uint32_t)Q;
        uint32_t br = (uint32_t)(S - 1) - a;
        uint32_t slot = logS ? (__brev(br) >> (32 - logS)) : 0u;
        return (int)(((uint32_t)(Q - 1) - bq) << logS | slot);
    }
};

__device__ __forceinline__ FpsRank make_rank(int n) {
    // opt_n_threads (cuda_utils.h:10-14): S = min(2^floor(log2 n), 1024)
    int lg = 31 - __clz(n);
    if (lg > 10) lg = 10;
    FpsRank r;
    r.logS = lg;
    r.S = 1 << lg;
    r.Q = (n + r.S - 1) >> lg;
    return r;
}

// One FPS pass over the n points held in LDS (sx/sy/sz) selecting m of them, by T threads (T / 64 waves) with PPT points each in registers.
// temp_io: optional global running-min buffer (API semantics) - read at start, written back at the end.
//
// A pick is one dependent chain, so its length in issue slots is the kernel's run time.  Per pick: the winner's coordinates (LDS), the
// distance update (packed f32, two points per instruction), the running minimum (v_min_u32 on the bit patterns), the thread's best
// (distance, rank) pair (one v_max_f64 per point: max_key), the wave's largest distance (six DPP steps), then the tie rule - the largest
// RANK among the lanes that hold that distance (six more DPP steps).  T == 64: the wave's result is the cloud's, in scalar registers, and the next pick
// starts at once; T > 64: one 64-bit LDS slot per wave, one barrier.
// Measured at 320 clouds of 1024 -> 512 -> 128 (round 5, profiles/r5_fps_waves.txt): the round-4 form (four waves x 4 points, compiler-
// scheduled reductions, per-point (distance, rank) tracking) 414 us; this pass on four / two / one wave(s): 271 / 279 / 246 us (5 clouds: 230
// either way - the exchange through LDS and the barrier cost what the extra points per lane cost).  One wave ships for n <= 1024.
template <int AR, int PPT, int T, bool FULL = false>  // FULL: n == PPT * T exactly (no bounds checks when the points are loaded)
__device__ void fps_pass(int n, int m, lds_cptr sx, lds_cptr sy, lds_cptr sz, float *temp_io,
                         int32_t *idx_out, unsigned long long (*slots)[FPS_T / 64]) {
    const int tid = threadIdx.x;
    const FpsRank rk = make_rank(n);
    float px[PPT], py[PPT], pz[PPT];
    u32x2 pt[PPT];  // per point: {rank (fixed), bits of the running minimum}: the 64-bit key of max_key, kept as a register pair
    // points past the end: running minimum +0 and rank 0 - min(d, 0) stays 0, so they are never ahead of a real point and the loop
    // needs no bounds checks (squared distances are >= +0 for finite clouds, where the unsigned order of the bits IS the float order)
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        int k = tid + j * T;
        bool ok = FULL || k < n;
        px[j] = ok ? sx[k] : 0.f;
        py[j] = ok ? sy[k] : 0.f;
        pz[j] = ok ? sz[k] : 0.f;
        pt[j].y = __float_as_uint(ok ? (temp_io ? temp_io[k] : 1e10f) : 0.f);
        pt[j].x = ok ? rk.rank(k) : 0u;
    }
    int old = 0;
    [[maybe_unused]] int mine = 0;  // T == 64: lane (it % 64) keeps pick it; flushed to idx_out every 64 picks
    if (T > 64 && tid == 0) idx_out[0] = 0;
    for (int it = 1; it < m; ++it) {
        float x1 = sx[old], y1 = sy[old], z1 = sz[old];
        float dist[PPT];
        if constexpr (PPT % 2 == 0) {
            // two points per instruction on the packed-f32 VALU (the same IEEE operations, in the same order, as sqdist<AR>)
            const f32x2 X1 = {x1, x1}, Y1 = {y1, y1}, Z1 = {z1, z1};
#pragma unroll
            for (int j = 0; j < PPT; j += 2) {
                const f32x2 dx = f32x2{px[j], px[j + 1]} - X1, dy = f32x2{py[j], py[j + 1]} - Y1, dz = f32x2{pz[j], pz[j + 1]} - Z1;
                const f32x2 d = sqnorm_pk<AR>(dx, dy, dz);
                dist[j] = d.x, dist[j + 1] = d.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < PPT; ++j) dist[j] = sqdist<AR>(px[j], py[j], pz[j], x1, y1, z1);
        }
        // fminf(d, tmp) on the bit patterns (one v_min_u32 instead of canonicalise + v_min_f32), then the thread's best (distance, rank) pair:
        // lexicographic max of PPT 64-bit keys, PPT - 1 instructions (max_key)
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const uint32_t du = __float_as_uint(dist[j]), tu = pt[j].y;
            pt[j].y = du < tu ? du : tu;  // (written straight into the high half of the point's pair)
        }
        u32x2 key[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) key[j] = pt[j];
#pragma unroll
        for (int st = 1; st < PPT; st <<= 1)
#pragma unroll
            for (int j = 0; j + st < PPT; j += 2 * st) key[j] = max_key(key[j], key[j + st]);
        const uint32_t bd = key[0].y, br = key[0].x;
        // the wave's: largest distance, then the largest rank among the lanes that hold it
        const uint32_t wm = wave_max_u32(bd);
        const uint32_t cr = bd == wm ? br : 0u;
        const uint32_t wr = wave_max_u32(cr);
        if constexpr (T == 64) {
            old = rk.unrank(wr);
            if ((it & 63) == 0) {  // picks it-64 .. it-1 are complete (pick 0 is point 0: `mine` starts at 0)
                idx_out[it - 64 + tid] = mine;
            }
            mine = tid == (it & 63) ? old : mine;
        } else {
            const int par = it & 1;
            if ((tid & 63) == 0) slots[par][tid >> 6] = ((unsigned long long)wm << 32) | wr;
            __syncthreads();
            unsigned long long v = slots[par][0];
#pragma unroll
            for (int w = 1; w < T / 64; ++w) {
                unsigned long long o = slots[par][w];
                v = o > v ? o : v;
            }
            old = rk.unrank((uint32_t)(v & 0xffffffffull));
            if (tid == 0) idx_out[it] = old;
        }
    }
    if constexpr (T == 64) {
        if (m > 0) {
            const int base = (m - 1) & ~63;
            if (base + tid < m) idx_out[base + tid] = mine;
        }
    }
    if (temp_io) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            int k = tid + j * T;
            if (k < n) temp_io[k] = __uint_as_float(pt[j].y);
        }
    }
}

// Fallback for n > FPS_T*FPS_MAXPPT: running min kept in global memory (temp must be provided).
template <int AR>
__device__ void fps_pass_big(int n, int m, const float *xyz, float *temp, int32_t *idx_out,
                             unsigned long long (*slots)[FPS_T / 64]) {
    const int tid = threadIdx.x;
    const FpsRank rk = make_rank(n);
    int old = 0;
    if (tid == 0) idx_out[0] = 0;
    for (int it = 1; it < m; ++it) {
        float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
        uint32_t bd = 0u, br = 0u;
        for (int k = tid; k < n; k += FPS_T) {
            float d = sqdist<AR>(xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2], x1, y1, z1);
            float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            const uint32_t dk = fkey(d2), rr = rk.rank(k);
            const bool better = dk > bd || (dk == bd && rr > br);
            bd = better ? dk : bd;
            br = better ? rr : br;
        }
        const uint32_t wm = wave_max_u32(bd);
        const uint32_t wr = wave_max_u32(bd == wm ? br : 0u);
        const int par = it & 1;
        if ((tid & 63) == 0) slots[par][tid >> 6] = ((unsigned long long)wm << 32) | wr;
        __syncthreads();
        unsigned long long v = slots[par][0];
#pragma unroll
        for (int w = 1; w < FPS_T / 64; ++w) {
            unsigned long long o = slots[par][w];
            v = o > v ? o : v;
        }
        old = rk.unrank((uint32_t)(v & 0xffffffffull));
        if (tid == 0) idx_out[it] = old;
    }
}

template <int AR, int PPT, int T>
__global__ __launch_bounds__(T) void fps_kernel(int n, int m, const float *__restrict__ xyz, float *__restrict__ temp,
                                                int32_t *__restrict__ idx) {
    extern __shared__ float lds[];  // sx[n] sy[n] sz[n]
    __shared__ unsigned long long slots[2][FPS_T / 64];
    const int b = blockIdx.x;
    xyz += (size_t)b * n * 3;
    temp += (size_t)b * n;
    idx += (size_t)b * m;
    float *sx = lds, *sy = lds + n, *sz = lds + 2 * n;
    for (int i = threadIdx.x; i < n * 3; i += T) {
        float v = xyz[i];
        int k = i / 3, c = i - k * 3;
        (c == 0 ? sx : c == 1 ? sy : sz)[k] = v;
    }
    __syncthreads();
    fps_pass<AR, PPT, T>(n, m, as_lds(sx), as_lds(sy), as_lds(sz), temp, idx, slots);
}

template <int AR>
__global__ __launch_bounds__(FPS_T) void fps_big_kernel(int n, int m, const float *__restrict__ xyz, float *__restrict__ temp,
                                                        int32_t *__restrict__ idx) {
    __shared__ unsigned long long slots[2][FPS_T / 64];
    const int b = blockIdx.x;
    fps_pass_big<AR>(n, m, xyz + (size_t)b * n * 3, temp + (size_t)b * n, idx + (size_t)b * m, slots);
}

// FPS + gather for up to three consecutive levels (encoder path): n0 <= 1024.
struct FpsChainArgs {
    int n0, nlevels;
    int m[3];
    const float *xyz;
    int32_t *idx[3];
    float *new_xyz[3];
};

// one level of the chain with the widest register-resident form that fits: n <= PPT * T
template <int AR, int T, int PPT>
__device__ __forceinline__ void fps_level(int n, int m, const float *sx, const float *sy, const float *sz, int32_t *sel,
                                          unsigned long long (*slots)[FPS_T / 64]) {
    if constexpr (PPT == 1) {
        if (n == T)
            fps_pass<AR, 1, T, true>(n, m, as_lds(sx), as_lds(sy), as_lds(sz), nullptr, sel, slots);
        else
            fps_pass<AR, 1, T>(n, m, as_lds(sx), as_lds(sy), as_lds(sz), nullptr, sel, slots);
    } else {
        if (n == PPT * T)
            fps_pass<AR, PPT, T, true>(n, m, as_lds(sx), as_lds(sy), as_lds(sz), nullptr, sel, slots);
        else if (n <= PPT * T / 2)
            fps_level<AR, T, PPT / 2>(n, m, sx, sy, sz, sel, slots);
        else
            fps_pass<AR, PPT, T>(n, m, as_lds(sx), as_lds(sy), as_lds(sz), nullptr, sel, slots);
    }
}

// LDS (dynamic): coordinate planes A [3][n0] and B [3][m0] that the levels ping-pong between (every level is at most as large as
// the one before, so level l+1 always fits the buffer level l-1 lived in) + the selected indices [m0]: 20.3 KB for 1024 -> 512 ->
// 256 -> 128 - small enough to share a CU with a 135 KB score-network workgroup of another stream.
template <int AR, int T>
__global__ __launch_bounds__(T) void fps_chain_kernel(FpsChainArgs a) {
    extern __shared__ float fps_lds[];
    __shared__ unsigned long long slots[2][FPS_T / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cap[2] = {a.n0, a.m[0]};
    float *plane[2] = {fps_lds, fps_lds + 3 * a.n0};
    int32_t *sel = reinterpret_cast<int32_t *>(fps_lds + 3 * a.n0 + 3 * a.m[0]);
    const float *xyz = a.xyz + (size_t)b * a.n0 * 3;
    for (int i = tid; i < a.n0 * 3; i += T) {
        int k = i / 3, c = i - k * 3;
        plane[0][c * cap[0] + k] = xyz[i];
    }
    __syncthreads();
    int n = a.n0, buf = 0;
    for (int l = 0; l < a.nlevels; ++l) {
        const int m = a.m[l];
        float *sx = plane[buf], *sy = plane[buf] + cap[buf], *sz = plane[buf] + 2 * cap[buf];
        float *nx = plane[buf ^ 1], *ny = plane[buf ^ 1] + cap[buf ^ 1], *nz = plane[buf ^ 1] + 2 * cap[buf ^ 1];
        fps_level<AR, T, 1024 / T>(n, m, sx, sy, sz, sel, slots);
        __syncthreads();
        int32_t *gi = a.idx[l] + (size_t)b * m;
        float *gx = a.new_xyz[l] + (size_t)b * m * 3;
        for (int j = tid; j < m; j += T) {
            int s = sel[j];
            float x = sx[s], y = sy[s], z = sz[s];
            gi[j] = s;
            gx[j * 3 + 0] = x;
            gx[j * 3 + 1] = y;
            gx[j * 3 + 2] = z;
            nx[j] = x;
            ny[j] = y;
            nz[j] = z;
        }
        __syncthreads();
        n = m;
        buf ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------- ball query
constexpr int BQ_T = 256;
// centres per wave.  Measured at 320 clouds (round 4): 8 / 16 centres per wave (fewer, longer workgroups that amortise the staging of the
// cloud) cost +25 %; two centres in flight per wave (shared candidate reads, independent ballot chains) +5 %: the kernel is bound by its
// many short waves' scalar chains, and more, shorter waves schedule best
constexpr int BQ_CPW = 4;

// LDS of a ball-query workgroup: the cloud as it lies in memory ([n][3]: lane k reads words 3k .. 3k+2 - stride 3 is coprime with the 64
// banks) + per wave a staging row per scale of nsample slots and one dump slot
__host__ __device__ constexpr size_t bq_lds_bytes(int n, int ns0, int ns1) {
    return ((size_t)n * 3 + (size_t)(BQ_T / 64) * (size_t)(ns0 + 1 + (ns1 > 0 ? ns1 + 1 : 0))) * 4;
}

// One wave scans the cloud in index order, 64 candidates per step.  NS = number of scales (1 or 2).
// The kernel is bound by instruction issue (a REAL275-shaped level-0 centre finds 55 / 188 of 1024 candidates inside its two radii and scans
// 8.9 of the 16 steps before both neighbourhoods are full), so a step is kept short: per scale one compare, v_mbcnt for the lane's slot, one
// LDS write of the hits into the wave's staging row (slots past nsample fall into the dump slot: no second predicate, no per-lane 64-bit
// global address, no scattered stores) and a scalar popcount; the row goes out with one coalesced store per centre, the first hit filling
// the empty slots (ball_query_gpu.cu:35-39).  Round 4's form (hits stored straight to global memory, first hit tracked with s_ff1,
// `if (mask)` / `if (count < nsample)` branches around each scale) took 85 instructions and 8 branches per step.
template <int AR, int NS, bool ZERO_FILL, bool FULL>  // FULL: n % 64 == 0 (no bounds checks in the scan)
__global__ __launch_bounds__(BQ_T) void ball_query_kernel(int n, int m, float r0, int ns0, float r1, int ns1,
                                                          const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                                                          int32_t *__restrict__ idx0, int32_t *__restrict__ idx1) {
    extern __shared__ float lds[];
    const int b = blockIdx.y;
    xyz += (size_t)b * n * 3;
    if ((n & 3) == 0) {  // 12 n bytes per cloud: 16-byte aligned rows
        const float4 *src = reinterpret_cast<const float4 *>(xyz);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < n * 3 / 4; i += BQ_T) dst[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < n * 3; i += BQ_T) lds[i] = xyz[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float rr0 = __fmul_rn(r0, r0), rr1 = __fmul_rn(r1, r1);  // ball_query_gpu.cu:23 (f32)
    int32_t *st0 = reinterpret_cast<int32_t *>(lds + (size_t)n * 3) + wave * (ns0 + 1 + (NS > 1 ? ns1 + 1 : 0));
    int32_t *st1 = st0 + ns0 + 1;
    for (int ci = 0; ci < BQ_CPW; ++ci) {
        const int p = (blockIdx.x * (BQ_T / 64) + wave) * BQ_CPW + ci;
        if (p >= m) break;
        const float *c = new_xyz + ((size_t)b * m + p) * 3;
        const float cx = c[0], cy = c[1], cz = c[2];
        int cnt0 = 0, cnt1 = 0;
        const float *cand = lds + 3 * lane;
        for (int base = 0; base < n; base += 64, cand += 3 * 64) {
            const int k = base + lane;
            const bool in = FULL || k < n;
            const float *q = in ? cand : lds;
            const float d2 = sqdist<AR>(cx, cy, cz, q[0], q[1], q[2]);
            {
                const bool hit = in && d2 < rr0;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(hit);
                const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, (uint32_t)cnt0));
                if (hit) st0[slot < ns0 ? slot : ns0] = k;
                cnt0 += __popcll(mk);
            }
            if constexpr (NS > 1) {
                const bool hit = in && d2 < rr1;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(hit);
                const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, (uint32_t)cnt1));
                if (hit) st1[slot < ns1 ? slot : ns1] = k;
                cnt1 += __popcll(mk);
            }
            if (cnt0 >= ns0 && (NS == 1 || cnt1 >= ns1)) break;
        }
        // a wave's LDS operations complete in order: the rows are read back by other lanes without a barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // hits in index order, then the first hit in every empty slot (ball_query_gpu.cu:35-39); no hit at all: untouched / zero
        if (cnt0 > 0 || ZERO_FILL) {
            int32_t *o0 = idx0 + ((size_t)b * m + p) * ns0;
            const int first = cnt0 > 0 ? st0[0] : 0;
            for (int sl = lane; sl < ns0; sl += 64) o0[sl] = sl < cnt0 ? st0[sl] : first;
        }
        if (NS > 1 && (cnt1 > 0 || ZERO_FILL)) {
            int32_t *o1 = idx1 + ((size_t)b * m + p) * ns1;
            const int first = cnt1 > 0 ? st1[0] : 0;
            for (int sl = lane; sl < ns1; sl += 64) o1[sl] = sl < cnt1 ? st1[sl] : first;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Large-n fallback (cloud does not fit LDS): candidates straight from global memory (L1/L2 resident).
template <int AR>
__global__ __launch_bounds__(BQ_T) void ball_query_big_kernel(int n, int m, float r0, int ns0, const float *__restrict__ new_xyz,
                                                              const float *__restrict__ xyz, int32_t *__restrict__ idx0) {
    const int b = blockIdx.y;
    xyz += (size_t)b * n * 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float rr0 = __fmul_rn(r0, r0);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int p = blockIdx.x * (BQ_T / 64) + wave;
    if (p >= m) return;
    const float *c = new_xyz + ((size_t)b * m + p) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    int32_t *o0 = idx0 + ((size_t)b * m + p) * ns0;
    int cnt0 = 0, first0 = 0;
    for (int base = 0; base < n && cnt0 < ns0; base += 64) {
        const int k = base + lane;
        float d2 = 3.0e38f;
        if (k < n) d2 = sqdist<AR>(cx, cy, cz, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]);
        unsigned long long mk = __ballot(k < n && d2 < rr0);
        if (mk) {
            if (cnt0 == 0) first0 = base + __ffsll((long long)mk) - 1;
            int slot = cnt0 + __popcll(mk & below);
            if (((mk >> lane) & 1ull) && slot < ns0) o0[slot] = k;
            cnt0 += __popcll(mk);
        }
    }
    if (cnt0 > 0)
        for (int sl = (cnt0 < ns0 ? cnt0 : ns0) + lane; sl < ns0; sl += 64) o0[sl] = first0;
}

// ---------------------------------------------------------------------------------------------- gather / group
__global__ void gather_points_kernel(int c, int n, int m, const float *__restrict__ points, const int32_t *__restrict__ idx,
                                     float *__restrict__ out) {
    const int b = blockIdx.z, ci = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    out[((size_t)b * c + ci) * m + p] = points[((size_t)b * c + ci) * n + idx[(size_t)b * m + p]];
}

__global__ void gather_points_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
                                          float *__restrict__ grad_points) {
    const int b = blockIdx.z, ci = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    atomicAdd(grad_points + ((size_t)b * c + ci) * n + idx[(size_t)b * m + p], grad_out[((size_t)b * c + ci) * m + p]);
}

__global__ void group_points_kernel(int c, int n, int q, const float *__restrict__ points, const int32_t *__restrict__ idx,
                                    float *__restrict__ out) {
    // q = npoints*nsample; one thread per (b, c, q)
    const int b = blockIdx.z, ci = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q) return;
    out[((size_t)b * c + ci) * q + i] = points[((size_t)b * c + ci) * n + idx[(size_t)b * q + i]];
}

__global__ void group_points_grad_kernel(int c, int n, int q, const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
                                         float *__restrict__ grad_points) {
    const int b = blockIdx.z, ci = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q) return;
    atomicAdd(grad_points + ((size_t)b * c + ci) * n + idx[(size_t)b * q + i], grad_out[((size_t)b * c + ci) * q + i]);
}

// ---------------------------------------------------------------------------------------------- three_nn / interpolate
template <int AR>
__global__ void three_nn_kernel(int n, int m, const float *__restrict__ unknown, const float *__restrict__ known,
                                float *__restrict__ dist2, int32_t *__restrict__ idx) {
    extern __shared__ float lds[];  // kx[m] ky[m] kz[m] (tiled in chunks of TILE)
    constexpr int TILE = 2048;
    const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    known += (size_t)b * m * 3;
    float ux = 0, uy = 0, uz = 0;
    if (p < n) {
        const float *u = unknown + ((size_t)b * n + p) * 3;
        ux = u[0], uy = u[1], uz = u[2];
    }
    // interpolate_gpu.cu:30: doubles initialised to 1e40, compared against the float distance
    double best1 = 1e40, best2 = 1e40, best3 = 1e40;
    int b1 = 0, b2 = 0, b3 = 0;
    float *kx = lds, *ky = lds + TILE, *kz = lds + 2 * TILE;
    for (int base = 0; base < m; base += TILE) {
        const int cnt = min(TILE, m - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 3; i += blockDim.x) {
            int k = i / 3, c = i - k * 3;
            (c == 0 ? kx : c == 1 ? ky : kz)[k] = known[(size_t)base * 3 + i];
        }
        __syncthreads();
        if (p < n) {
            for (int k = 0; k < cnt; ++k) {
                double d = (double)sqdist<AR>(ux, uy, uz, kx[k], ky[k], kz[k]);
                int kk = base + k;
                if (d < best1) {
                    best3 = best2; b3 = b2;
                    best2 = best1; b2 = b1;
                    best1 = d; b1 = kk;
                } else if (d < best2) {
                    best3 = best2; b3 = b2;
                    best2 = d; b2 = kk;
                } else if (d < best3) {
                    best3 = d; b3 = kk;
                }
            }
        }
    }
    if (p < n) {
        float *o = dist2 + ((size_t)b * n + p) * 3;
        int32_t *oi = idx + ((size_t)b * n + p) * 3;
        o[0] = (float)best1; o[1] = (float)best2; o[2] = (float)best3;
        oi[0] = b1; oi[1] = b2; oi[2] = b3;
    }
}

template <int AR>
__global__ void three_interpolate_kernel(int c, int m, int n, const float *__restrict__ points, const int32_t *__restrict__ idx,
                                         const float *__restrict__ weight, float *__restrict__ out) {
    const int b = blockIdx.z, ci = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float *w = weight + ((size_t)b * n + p) * 3;
    const int32_t *id = idx + ((size_t)b * n + p) * 3;
    const float *src = points + ((size_t)b * c + ci) * m;
    // w0*p0 + w1*p1 + w2*p2 (interpolate_gpu.cu:95): the same three-product sum, the same conventions
    out[((size_t)b * c + ci) * n + p] = dot3<AR>(w[0], src[id[0]], w[1], src[id[1]], w[2], src[id[2]]);
}

__global__ void three_interpolate_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
                                              const float *__restrict__ weight, float *__restrict__ grad_points) {
    const int b = blockIdx.z, ci = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float g = grad_out[((size_t)b * c + ci) * n + p];
    const float *w = weight + ((size_t)b * n + p) * 3;
    const int32_t *id = idx + ((size_t)b * n + p) * 3;
    float *gp = grad_points + ((size_t)b * c + ci) * m;
    atomicAdd(gp + id[0], g * w[0]);
    atomicAdd(gp + id[1], g * w[1]);
    atomicAdd(gp + id[2], g * w[2]);
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

// run `stmt` with AR = the compile-time constant of `arith` (validated by the caller)
#define GP_ARITH_SWITCH(arith, stmt)                                              \
    switch (arith) {                                                              \
        case GP_ARITH_A: { constexpr int AR = GP_ARITH_A; stmt; } break;          \
        case GP_ARITH_B: { constexpr int AR = GP_ARITH_B; stmt; } break;          \
        default: { constexpr int AR = GP_ARITH_C; stmt; } break;                  \
    }

static bool arith_ok(int arith) { return arith == GP_ARITH_A || arith == GP_ARITH_B || arith == GP_ARITH_C; }

int gp_arith_default(void) { return GP_ARITH_DEFAULT; }

int gp_furthest_point_sampling_arith(int arith, int b, int n, int m, const float *xyz, float *temp, int32_t *idx, gp_stream_t s) {
    if (!arith_ok(arith) || b < 0 || n <= 0 || m < 0 || !xyz || !temp || !idx) return GP_EINVAL;
    if (b == 0 || m == 0) return GP_OK;
    if (m > n) return GP_EINVAL;
    hipStream_t st = (hipStream_t)s;
    const size_t lds = (size_t)n * 3 * sizeof(float);
    GP_ARITH_SWITCH(arith, {
        if (n <= FPS_W)
            hipLaunchKernelGGL((fps_kernel<AR, 1, FPS_W>), dim3(b), dim3(FPS_W), lds, st, n, m, xyz, temp, idx);
        else if (n <= 2 * FPS_W)
            hipLaunchKernelGGL((fps_kernel<AR, 2, FPS_W>), dim3(b), dim3(FPS_W), lds, st, n, m, xyz, temp, idx);
        else if (n <= 4 * FPS_W)
            hipLaunchKernelGGL((fps_kernel<AR, 4, FPS_W>), dim3(b), dim3(FPS_W), lds, st, n, m, xyz, temp, idx);
        else if (n <= 8 * FPS_W)
            hipLaunchKernelGGL((fps_kernel<AR, 8, FPS_W>), dim3(b), dim3(FPS_W), lds, st, n, m, xyz, temp, idx);
        else if (n <= 16 * FPS_W)
            hipLaunchKernelGGL((fps_kernel<AR, 16, FPS_W>), dim3(b), dim3(FPS_W), lds, st, n, m, xyz, temp, idx);
        else if (n <= 8 * FPS_T)
            hipLaunchKernelGGL((fps_kernel<AR, 8, FPS_T>), dim3(b), dim3(FPS_T), lds, st, n, m, xyz, temp, idx);
        else if (n <= FPS_MAXPPT * FPS_T)
            hipLaunchKernelGGL((fps_kernel<AR, FPS_MAXPPT, FPS_T>), dim3(b), dim3(FPS_T), lds, st, n, m, xyz, temp, idx);
        else
            hipLaunchKernelGGL((fps_big_kernel<AR>), dim3(b), dim3(FPS_T), 0, st, n, m, xyz, temp, idx);
    })
    return gp_launch_status();
}
int gp_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int32_t *idx, gp_stream_t s) {
    return gp_furthest_point_sampling_arith(GP_ARITH_DEFAULT, b, n, m, xyz, temp, idx, s);
}

int gp_fps_chain_arith(int arith, int b, int n0, int nlevels, const int *m, const float *xyz, int32_t *idx0, float *new_xyz0, int32_t *idx1,
                       float *new_xyz1, int32_t *idx2, float *new_xyz2, gp_stream_t s) {
    if (!arith_ok(arith) || b < 0 || n0 <= 0 || n0 > 1024 || nlevels < 1 || nlevels > 3 || !m || !xyz) return GP_EINVAL;
    if (b == 0) return GP_OK;
    FpsChainArgs a;
    a.n0 = n0;
    a.nlevels = nlevels;
    a.xyz = xyz;
    int32_t *ii[3] = {idx0, idx1, idx2};
    float *xx[3] = {new_xyz0, new_xyz1, new_xyz2};
    int prev = n0;
    for (int l = 0; l < 3; ++l) {
        a.m[l] = l < nlevels ? m[l] : 0;
        a.idx[l] = ii[l];
        a.new_xyz[l] = xx[l];
        if (l < nlevels) {
            if (m[l] <= 0 || m[l] > prev || !ii[l] || !xx[l]) return GP_EINVAL;
            prev = m[l];
        }
    }
    const size_t lds = ((size_t)3 * n0 + 4 * (size_t)a.m[0]) * sizeof(float);
    GP_ARITH_SWITCH(arith, hipLaunchKernelGGL((fps_chain_kernel<AR, FPS_W>), dim3(b), dim3(FPS_W), lds, (hipStream_t)s, a))
    return gp_launch_status();
}
int gp_fps_chain(int b, int n0, int nlevels, const int *m, const float *xyz, int32_t *idx0, float *new_xyz0, int32_t *idx1,
                 float *new_xyz1, int32_t *idx2, float *new_xyz2, gp_stream_t s) {
    return gp_fps_chain_arith(GP_ARITH_DEFAULT, b, n0, nlevels, m, xyz, idx0, new_xyz0, idx1, new_xyz1, idx2, new_xyz2, s);
}

int gp_ball_query_arith(int arith, int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int32_t *idx,
                        gp_stream_t s) {
    if (!arith_ok(arith) || b < 0 || n <= 0 || m < 0 || nsample <= 0 || !new_xyz || !xyz || !idx) return GP_EINVAL;
    if (b == 0 || m == 0) return GP_OK;
    hipStream_t st = (hipStream_t)s;
    if (bq_lds_bytes(n, nsample, 0) <= 60 * 1024) {
        dim3 grid((m + (BQ_T / 64) * BQ_CPW - 1) / ((BQ_T / 64) * BQ_CPW), b);
        GP_ARITH_SWITCH(arith, {
            if (n % 64 == 0)
                hipLaunchKernelGGL((ball_query_kernel<AR, 1, false, true>), grid, dim3(BQ_T), bq_lds_bytes(n, nsample, 0), st, n, m, radius, nsample, 0.f, 0,
                                   new_xyz, xyz, idx, (int32_t *)nullptr);
            else
                hipLaunchKernelGGL((ball_query_kernel<AR, 1, false, false>), grid, dim3(BQ_T), bq_lds_bytes(n, nsample, 0), st, n, m, radius, nsample, 0.f, 0,
                                   new_xyz, xyz, idx, (int32_t *)nullptr);
        })
    } else {
        dim3 grid((m + BQ_T / 64 - 1) / (BQ_T / 64), b);
        GP_ARITH_SWITCH(arith, hipLaunchKernelGGL((ball_query_big_kernel<AR>), grid, dim3(BQ_T), 0, st, n, m, radius, nsample, new_xyz, xyz, idx))
    }
    return gp_launch_status();
}
int gp_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int32_t *idx, gp_stream_t s) {
    return gp_ball_query_arith(GP_ARITH_DEFAULT, b, n, m, radius, nsample, new_xyz, xyz, idx, s);
}

int gp_ball_query_msg_arith(int arith, int b, int n, int m, float radius0, int nsample0, float radius1, int nsample1, const float *new_xyz,
                            const float *xyz, int32_t *idx0, int32_t *idx1, gp_stream_t s) {
    if (!arith_ok(arith) || b < 0 || n <= 0 || m < 0 || nsample0 <= 0 || nsample1 <= 0 || !new_xyz || !xyz || !idx0 || !idx1) return GP_EINVAL;
    if (bq_lds_bytes(n, nsample0, nsample1) > 60 * 1024) return GP_EINVAL;
    if (b == 0 || m == 0) return GP_OK;
    dim3 grid((m + (BQ_T / 64) * BQ_CPW - 1) / ((BQ_T / 64) * BQ_CPW), b);
    GP_ARITH_SWITCH(arith, {
        if (n % 64 == 0)
            hipLaunchKernelGGL((ball_query_kernel<AR, 2, true, true>), grid, dim3(BQ_T), bq_lds_bytes(n, nsample0, nsample1), (hipStream_t)s, n, m, radius0,
                               nsample0, radius1, nsample1, new_xyz, xyz, idx0, idx1);
        else
            hipLaunchKernelGGL((ball_query_kernel<AR, 2, true, false>), grid, dim3(BQ_T), bq_lds_bytes(n, nsample0, nsample1), (hipStream_t)s, n, m, radius0,
                               nsample0, radius1, nsample1, new_xyz, xyz, idx0, idx1);
    })
    return gp_launch_status();
}
int gp_ball_query_msg(int b, int n, int m, float radius0, int nsample0, float radius1, int nsample1, const float *new_xyz,
                      const float *xyz, int32_t *idx0, int32_t *idx1, gp_stream_t s) {
    return gp_ball_query_msg_arith(GP_ARITH_DEFAULT, b, n, m, radius0, nsample0, radius1, nsample1, new_xyz, xyz, idx0, idx1, s);
}

int gp_gather_points(int b, int c, int n, int m, const float *points, const int32_t *idx, float *out, gp_stream_t s) {
    if (b < 0 || c < 0 || n <= 0 || m < 0 || !points || !idx || !out) return GP_EINVAL;
    if (b == 0 || c == 0 || m == 0) return GP_OK;
    if (c > 65535 || b > 65535) return GP_EINVAL;
    hipLaunchKernelGGL(gather_points_kernel, dim3((m + 255) / 256, c, b), dim3(256), 0, (hipStream_t)s, c, n, m, points, idx, out);
    return gp_launch_status();
}

int gp_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx, float *grad_points, gp_stream_t s) {
    if (b < 0 || c < 0 || n <= 0 || m < 0 || !grad_out || !idx || !grad_points) return GP_EINVAL;
    if (b == 0 || c == 0 || m == 0) return GP_OK;
    if (c > 65535 || b > 65535) return GP_EINVAL;
    hipLaunchKernelGGL(gather_points_grad_kernel, dim3((m + 255) / 256, c, b), dim3(256), 0, (hipStream_t)s, c, n, m, grad_out, idx,
                       grad_points);
    return gp_launch_status();
}

int gp_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int32_t *idx, float *out, gp_stream_t s) {
    if (b < 0 || c < 0 || n <= 0 || npoints < 0 || nsample < 0 || !points || !idx || !out) return GP_EINVAL;
    const int q = npoints * nsample;
    if (b == 0 || c == 0 || q == 0) return GP_OK;
    if (c > 65535 || b > 65535) return GP_EINVAL;
    hipLaunchKernelGGL(group_points_kernel, dim3((q + 255) / 256, c, b), dim3(256), 0, (hipStream_t)s, c, n, q, points, idx, out);
    return gp_launch_status();
}

int gp_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int32_t *idx, float *grad_points,
                         gp_stream_t s) {
    if (b < 0 || c < 0 || n <= 0 || npoints < 0 || nsample < 0 || !grad_out || !idx || !grad_points) return GP_EINVAL;
    const int q = npoints * nsample;
    if (b == 0 || c == 0 || q == 0) return GP_OK;
    if (c > 65535 || b > 65535) return GP_EINVAL;
    hipLaunchKernelGGL(group_points_grad_kernel, dim3((q + 255) / 256, c, b), dim3(256), 0, (hipStream_t)s, c, n, q, grad_out, idx,
                       grad_points);
    return gp_launch_status();
}

int gp_three_nn_arith(int arith, int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx, gp_stream_t s) {
    if (!arith_ok(arith) || b < 0 || n < 0 || m <= 0 || !unknown || !known || !dist2 || !idx) return GP_EINVAL;
    if (b == 0 || n == 0) return GP_OK;
    GP_ARITH_SWITCH(arith, hipLaunchKernelGGL((three_nn_kernel<AR>), dim3((n + 255) / 256, b), dim3(256), 3 * 2048 * sizeof(float), (hipStream_t)s,
                                              n, m, unknown, known, dist2, idx))
    return gp_launch_status();
}
int gp_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx, gp_stream_t s) {
    return gp_three_nn_arith(GP_ARITH_DEFAULT, b, n, m, unknown, known, dist2, idx, s);
}

int gp_three_interpolate_arith(int arith, int b, int c, int m, int n, const float *points, const int32_t *idx, const float *weight, float *out,
                               gp_stream_t s) {
    if (!arith_ok(arith) || b < 0 || c < 0 || m <= 0 || n < 0 || !points || !idx || !weight || !out) return GP_EINVAL;
    if (b == 0 || c == 0 || n == 0) return GP_OK;
    if (c > 65535 || b > 65535) return GP_EINVAL;
    GP_ARITH_SWITCH(arith, hipLaunchKernelGGL((three_interpolate_kernel<AR>), dim3((n + 255) / 256, c, b), dim3(256), 0, (hipStream_t)s, c, m, n,
                                              points, idx, weight, out))
    return gp_launch_status();
}
int gp_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx, const float *weight, float *out,
                         gp_stream_t s) {
    return gp_three_interpolate_arith(GP_ARITH_DEFAULT, b, c, m, n, points, idx, weight, out, s);
}

int gp_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx, const float *weight,
                              float *grad_points, gp_stream_t s) {
    if (b < 0 || c < 0 || m <= 0 || n < 0 || !grad_out || !idx || !weight || !grad_points) return GP_EINVAL;
    if (b == 0 || c == 0 || n == 0) return GP_OK;
    if (c > 65535 || b > 65535) return GP_EINVAL;
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3((n + 255) / 256, c, b), dim3(256), 0, (hipStream_t)s, c, n, m, grad_out, idx,
                       weight, grad_points);
    return gp_launch_status();
}

}  // extern "C"
