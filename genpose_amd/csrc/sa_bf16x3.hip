// OPT-IN, EXPLORATORY (round 5): one set-abstraction scale (hoisted first layer -> two dense layers -> max over the neighbourhood) with the
// dense layers on the BF16 matrix pipe as THREE-TERM SPLIT PRODUCTS, fp32 accumulate:
//       a . b  ~=  a_hi . b_hi  +  a_lo . b_hi  +  a_hi . b_lo ,      x_hi = bf16(x),  x_lo = bf16(x - x_hi)
// (relative error ~2^-17 per product - the dropped a_lo . b_lo term and the 8-bit lo parts - against 2^-24 for the fp32 pipe; the reference's own
// convolutions run at TF32 = 2^-11 on its stated RTX 3090 / PyTorch 1.12).  v_mfma_f32_16x16x32_bf16 issues 16x the FLOPs per cycle of
// v_mfma_f32_16x16x4_f32, so three of them per product are 5.3x the fp32 rate on paper.  NOT the default: the headline numbers and every parity
// claim are the fp32 kernels' (sa_mlp.hip); this file is reached only through gp_sa_pre_mlp_max_bf16x3 (encoder precision 'bf16x3').
// Furthest point sampling and the ball queries see coordinates only: centres and neighbourhoods are bit-identical either way.
//
// Form: sa_chain_ring_kernel's (sa_mlp.hip) - 8 waves per workgroup, activations register-resident from the gather to the pooled output,
// layer-2 weights LDS-resident, layer-3 weights through a 3-slot LDS ring shared by the waves, one barrier per ring step - with
//   * 32 rows per wave and iteration (two 16-row sub-chunks: one neighbourhood of 32, or two of 16), so that every weight fragment read
//     from LDS feeds two rows' worth of matrix instructions (at 16 rows the kernel would be bound by the LDS reads: the matrix work shrinks
//     5x, the fragment bytes do not);
//   * the D fragment of a layer (lane = point, four consecutive channels of a 16-channel chunk) still IS the next layer's operand: two chunks
//     (2m, 2m+1) make the eight k-values a lane holds for k-block m of v_mfma_f32_16x16x32_bf16 - the host packs the weights in that k order
//     (weights.py: pack_bf16x3);
//   * layer 3 in halves of eight output chunks (64 accumulator registers for the two sub-chunks), ring slice = (half, k-block) = 16 KB;
//   * level 1 (64-64/96-128: all weights fit LDS) runs the same kernel without the ring and without a barrier in the loop.
#include "bf16x3.h"

namespace {

using namespace gp_bf16x3;

struct SABfArgs {
    int n, np, zstride, zoff;
    const float *xyz, *new_xyz, *z;
    const int32_t *idx;
    const float *wxyz, *b1;  // layer 1 stays on the fp32 VALU (three FMAs per channel on top of the hoisted feature half)
    const bf16x8 *w2;        // [KB1][NC2][2 = hi, lo][64 lanes]
    const float *b2;         // [32 * KB2] (zero padded)
    const bf16x8 *w3;        // [2 halves][KB2][8 chunks][2 = hi, lo][64 lanes]
    const float *b3;
    float *out;
    int cout_total, cout_off;
};

// RING: layer-3 weights through the 3-slot ring (level 2: 224 KB of them); !RING: they are LDS-resident too and the loop has no barrier
// (level 1: 64-64/96-128, 32-48 KB) - sa_chain_lds_kernel's form.
template <int C1, int C2, int C3, int NS, bool RING>
__global__ __launch_bounds__(512) void sa_chain_bf16x3_kernel(SABfArgs a, int nunits_total) {
    constexpr int KB1 = C1 / 32, NC2 = (C2 + 15) / 16, KB2 = (NC2 + 1) / 2, NC3H = 8, NHALF = C3 / 128, NWV = 8, NTH = 512;
    static_assert(C1 % 32 == 0 && C3 % 128 == 0 && (NS == 16 || NS == 32), "grouping levels 1 and 2 of the light / dense / lighter encoders");
    constexpr int SLICE = NC3H * 2 * 64;   // bf16x8 (16 B) per slice: one (half, k-block) of layer 3 = 16 KB
    constexpr int PER_T = SLICE / NTH;
    constexpr int NSL = NHALF * KB2;       // slices per iteration
    extern __shared__ __attribute__((aligned(16))) float lds[];
    bf16x8 *w2l = reinterpret_cast<bf16x8 *>(lds);                 // [KB1][NC2][2][64] resident
    bf16x8 *ring = w2l + KB1 * NC2 * 2 * 64;                        // RING: [3][SLICE]; else all [NSL][SLICE] slices
    f32x4 *w1l = reinterpret_cast<f32x4 *>(ring + (RING ? 3 : NSL) * SLICE);  // [C1] rows (wx, wy, wz, b1)
    float *b2l = reinterpret_cast<float *>(w1l + C1);               // [32 KB2]
    float *b3l = b2l + 32 * KB2;                                    // [C3]
    const int tid = threadIdx.x, lane = tid & 63, pt = lane & 15, g = lane >> 4;
    for (int e = tid; e < KB1 * NC2 * 2 * 64; e += NTH) w2l[e] = a.w2[e];
    for (int e = tid; e < C1; e += NTH) {
        f32x4 w = *reinterpret_cast<const f32x4 *>(a.wxyz + e * 4);
        w.w = a.b1[e];
        w1l[e] = w;
    }
    for (int e = tid; e < 32 * KB2; e += NTH) b2l[e] = a.b2[e];
    for (int e = tid; e < C3; e += NTH) b3l[e] = a.b3[e];
    // ring prologue: slices 0 and 1 into slots 0 and 1; slice 2 held in registers
    bf16x8 hold[PER_T];
    if constexpr (RING) {
#pragma unroll
        for (int u = 0; u < PER_T; ++u) {
            ring[0 * SLICE + tid + u * NTH] = a.w3[0 * SLICE + tid + u * NTH];
            ring[1 * SLICE + tid + u * NTH] = a.w3[1 * SLICE + tid + u * NTH];
            hold[u] = a.w3[2 * SLICE + tid + u * NTH];
        }
    } else {
        for (int e = tid; e < NSL * SLICE; e += NTH) ring[e] = a.w3[e];
    }
    __syncthreads();
    const int wave_global = blockIdx.x * NWV + __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = gridDim.x * NWV;  // (scalar, as in the fp32 chain kernels)
    const int my_units = wave_global < nunits_total ? (nunits_total - wave_global + nwaves - 1) / nwaves : 0;
    // RING: every wave runs the same number of iterations (idle ones compute on clamped rows and store nothing): barrier counts match
    const int nits_wg = RING ? (nunits_total + nwaves - 1) / nwaves : my_units;
    // unit `it` of this wave: 32 consecutive (centre, sample) rows = sub-chunks s = 0, 1 of 16 rows
    auto unit_of = [&](int it) { return it < my_units ? wave_global + it * nwaves : 0; };
    auto load_idx = [&](int it, int (&j)[2]) {
        const size_t r0 = (size_t)unit_of(it) * 32;
#pragma unroll
        for (int s = 0; s < 2; ++s) j[s] = a.idx[r0 + 16 * s + pt];
    };
    auto centre_of = [&](int it, int s) { return (unit_of(it) * 32 + 16 * s) / NS; };
    auto load_d = [&](int it, const int (&j)[2], float (&d)[2][3]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cc = centre_of(it, s), bcl = cc / a.np;
            const float *xyz = a.xyz + (size_t)bcl * a.n * 3;
            const float *cp = a.new_xyz + (size_t)cc * 3;
            d[s][0] = xyz[j[s] * 3 + 0] - cp[0];  // grouped_xyz -= new_xyz (pointnet2_utils.py:253)
            d[s][1] = xyz[j[s] * 3 + 1] - cp[1];
            d[s][2] = xyz[j[s] * 3 + 2] - cp[2];
        }
    };
    int jcur[2], jn[2];
    float dcur[2][3];
    load_idx(0, jcur);
    load_d(0, jcur, dcur);
    load_idx(1, jn);
    int gstep = 0;  // global ring step: slice gstep % NSL sits in slot gstep % 3
#pragma unroll 1
    for (int it = 0; it < nits_wg; ++it) {
        int lo_ = lane;
        asm volatile("" : "+v"(lo_));  // keep the LDS weight reads inside the loop (see sa_chain_lds_kernel)
        const int g4 = (lo_ >> 4) * 4;
        // ---- layers 1 + 2: k-block by k-block; the hoisted feature rows of k-block kb + 1 are requested while kb is multiplied
        const float *zrow[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int bcl = centre_of(it, s) / a.np;
            zrow[s] = a.z + ((size_t)bcl * a.n + jcur[s]) * a.zstride + a.zoff + g4;
        }
        f32x4 zz[2][2][2];  // [buffer][sub][chunk of the k-block]
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < 2; ++c) zz[0][s][c] = *reinterpret_cast<const f32x4 *>(zrow[s] + 16 * c);
        f32x4 acc2[NC2][2];
#pragma unroll
        for (int n = 0; n < NC2; ++n) acc2[n][0] = acc2[n][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) {
            if (kb + 1 < KB1) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int c = 0; c < 2; ++c) zz[(kb + 1) & 1][s][c] = *reinterpret_cast<const f32x4 *>(zrow[s] + 32 * (kb + 1) + 16 * c);
            }
            bf16x8 h1hi[2], h1lo[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float dx = dcur[s][0], dy = dcur[s][1], dz = dcur[s][2];
                f32x4 h[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int q = 2 * kb + c;
                    const f32x4 r0 = w1l[16 * q + g4 + 0], r1 = w1l[16 * q + g4 + 1], r2 = w1l[16 * q + g4 + 2], r3 = w1l[16 * q + g4 + 3];
                    f32x4 v = zz[kb & 1][s][c];
                    v.x += (r0.x * dx + r0.y * dy + r0.z * dz) + r0.w;  // the fp32 kernels' layer-1 arithmetic
                    v.y += (r1.x * dx + r1.y * dy + r1.z * dz) + r1.w;
                    v.z += (r2.x * dx + r2.y * dy + r2.z * dz) + r2.w;
                    v.w += (r3.x * dx + r3.y * dy + r3.z * dz) + r3.w;
                    h[c] = relu4(v);
                }
                split8(h[0], h[1], h1hi[s], h1lo[s]);
            }
            // two output chunks at a time: four independent accumulators, the three terms in separate passes over them
#pragma unroll
            for (int n0 = 0; n0 < NC2; n0 += 2) {
                bf16x8 wh[2], wl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n = n0 + u < NC2 ? n0 + u : NC2 - 1;
                    wh[u] = w2l[((kb * NC2 + n) * 2 + 0) * 64 + lo_];
                    wl[u] = w2l[((kb * NC2 + n) * 2 + 1) * 64 + lo_];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (n0 + u < NC2) {
#pragma unroll
                        for (int s = 0; s < 2; ++s) acc2[n0 + u][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[u], h1hi[s], acc2[n0 + u][s], 0, 0, 0);
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (n0 + u < NC2) {
#pragma unroll
                        for (int s = 0; s < 2; ++s) acc2[n0 + u][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[u], h1hi[s], acc2[n0 + u][s], 0, 0, 0);
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (n0 + u < NC2) {
#pragma unroll
                        for (int s = 0; s < 2; ++s) acc2[n0 + u][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[u], h1lo[s], acc2[n0 + u][s], 0, 0, 0);
                    }
            }
        }
        // the next unit's indices were requested an iteration ago: its coordinates now, the indices of the one after
        const int cur_valid = it < my_units;
        const int unit = unit_of(it);
        load_d(it + 1, jn, dcur);
        jcur[0] = jn[0], jcur[1] = jn[1];
        load_idx(it + 2, jn);
        // ---- bias + ReLU + split of the hidden layer: k-block m of layer 3 = chunks 2m, 2m+1 (a missing odd chunk is zero)
        bf16x8 h2hi[KB2][2], h2lo[KB2][2];
#pragma unroll
        for (int m = 0; m < KB2; ++m)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 v0 = relu4(acc2[2 * m][s] + *reinterpret_cast<const f32x4 *>(b2l + 16 * (2 * m) + g4));
                f32x4 v1 = {0.f, 0.f, 0.f, 0.f};
                if (2 * m + 1 < NC2) v1 = relu4(acc2[2 * m + 1][s] + *reinterpret_cast<const f32x4 *>(b2l + 16 * (2 * m + 1) + g4));
                split8(v0, v1, h2hi[m][s], h2lo[m][s]);
            }
        // ---- layer 3 (transposed: lane = channel, registers x lane groups = the 16 points of a sub-chunk), eight output chunks at a time
#pragma unroll 1
        for (int half = 0; half < NHALF; ++half) {
            f32x4 acc3[NC3H][2];
#pragma unroll
            for (int n = 0; n < NC3H; ++n) acc3[n][0] = acc3[n][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB2; ++kb) {
                // slice gstep + 2 (held in registers since the previous step) -> its slot, last read in step gstep - 1
                if constexpr (RING) {
                    bf16x8 *dst = ring + ((gstep + 2) % 3) * SLICE;
#pragma unroll
                    for (int u = 0; u < PER_T; ++u) dst[tid + u * NTH] = hold[u];
                    const bf16x8 *src = a.w3 + (size_t)((gstep + 3) % NSL) * SLICE;
#pragma unroll
                    for (int u = 0; u < PER_T; ++u) hold[u] = src[tid + u * NTH];
                }
                const bf16x8 *slot = RING ? ring + (gstep % 3) * SLICE : ring + (half * KB2 + kb) * SLICE;
#pragma unroll
                for (int n0 = 0; n0 < NC3H; n0 += 2) {
                    bf16x8 wh[2], wl[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        wh[u] = slot[((n0 + u) * 2 + 0) * 64 + lo_];
                        wl[u] = slot[((n0 + u) * 2 + 1) * 64 + lo_];
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int s = 0; s < 2; ++s) acc3[n0 + u][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h2hi[kb][s], wh[u], acc3[n0 + u][s], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int s = 0; s < 2; ++s) acc3[n0 + u][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h2lo[kb][s], wh[u], acc3[n0 + u][s], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int s = 0; s < 2; ++s) acc3[n0 + u][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h2hi[kb][s], wl[u], acc3[n0 + u][s], 0, 0, 0);
                }
                if constexpr (RING) {
                    ++gstep;
                    __syncthreads();
                }
            }
            // pooling over the points (max_i relu(x_i + b) = relu(max_i x_i + b): bias and ReLU once per channel, after the pooling)
#pragma unroll
            for (int n = 0; n < NC3H; ++n) {
                const float m0 = points16_max_t(acc3[n][0]), m1 = points16_max_t(acc3[n][1]);
                const int ch = 16 * (half * NC3H + n) + (lo_ & 15);
                const float b = b3l[ch];
                if (cur_valid && g == 0) {
                    if (NS == 32) {
                        a.out[(size_t)unit * a.cout_total + a.cout_off + ch] = fmaxf(fmaxf(m0, m1) + b, 0.f);
                    } else {
                        a.out[(size_t)(2 * unit) * a.cout_total + a.cout_off + ch] = fmaxf(m0 + b, 0.f);
                        a.out[(size_t)(2 * unit + 1) * a.cout_total + a.cout_off + ch] = fmaxf(m1 + b, 0.f);
                    }
                }
            }
        }
    }
}

template <int C1, int C2, int C3, int NS, bool RING>
int launch_bf16x3(const SABfArgs &a, int b, hipStream_t st) {
    constexpr int KB1 = C1 / 32, NC2 = (C2 + 15) / 16, KB2 = (NC2 + 1) / 2, NSL = (C3 / 128) * KB2;
    const size_t lds = ((size_t)KB1 * NC2 * 2 * 64 + (RING ? 3 : NSL) * 8 * 2 * 64 + C1) * 16 + (size_t)(32 * KB2 + C3) * sizeof(float);
    if (lds > 160 * 1024) return GP_EINVAL;
    auto kern = sa_chain_bf16x3_kernel<C1, C2, C3, NS, RING>;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GP_ELAUNCH;
        done = true;
    }
    const int nunits = (int)(((size_t)b * a.np * NS) / 32);
    int blocks = (nunits + 7) / 8;
    const int per_cu = RING ? 1 : 2;  // persistent; the resident form (<= 90 KB of LDS, <= 128 registers for the 64-wide hidden layer) fits twice
    if (blocks > gp_num_cus() * per_cu) blocks = gp_num_cus() * per_cu;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, a, nunits);
    return gp_launch_status();
}

}  // namespace

extern "C" {

int gp_sa_pre_mlp_max_bf16x3(int b, int n, int np, int ns, int c1, int c2, int c3, const float *xyz, const float *new_xyz, const int32_t *idx,
                             const float *z, int zstride, int zoff, const float *wxyz, const float *bias1, const void *w2_split, const float *bias2,
                             const void *w3_split, const float *bias3, float *out, int cout_total, int cout_off, gp_stream_t s) {
    if (b < 0 || n <= 0 || np <= 0 || !xyz || !new_xyz || !idx || !z || !wxyz || !bias1 || !w2_split || !bias2 || !w3_split || !bias3 || !out)
        return GP_EINVAL;
    if ((cout_total & 3) || (cout_off & 3) || cout_off + c3 > cout_total || (zstride & 3) || (zoff & 3) || zoff + c1 > zstride) return GP_EINVAL;
    if (((size_t)b * np * ns) % 32) return GP_EINVAL;  // whole 32-row units
    if (b == 0) return GP_OK;
    SABfArgs a{n, np, zstride, zoff, xyz, new_xyz, z, idx, wxyz, bias1, reinterpret_cast<const bf16x8 *>(w2_split), bias2,
               reinterpret_cast<const bf16x8 *>(w3_split), bias3, out, cout_total, cout_off};
    if (c1 == 128 && c2 == 196 && c3 == 256 && ns == 32) return launch_bf16x3<128, 196, 256, 32, true>(a, b, (hipStream_t)s);
    if (c1 == 128 && c2 == 196 && c3 == 256 && ns == 16) return launch_bf16x3<128, 196, 256, 16, true>(a, b, (hipStream_t)s);
    if (c1 == 64 && c2 == 64 && c3 == 128 && ns == 16) return launch_bf16x3<64, 64, 128, 16, false>(a, b, (hipStream_t)s);
    if (c1 == 64 && c2 == 96 && c3 == 128 && ns == 32) return launch_bf16x3<64, 96, 128, 32, false>(a, b, (hipStream_t)s);
    if (c1 == 64 && c2 == 64 && c3 == 128 && ns == 32) return launch_bf16x3<64, 64, 128, 32, false>(a, b, (hipStream_t)s);
    return GP_EINVAL;  // grouping levels 1 and 2 of the light / dense / lighter encoders (exploratory)
}

}  // extern "C"
