// Forward + vector-Jacobian pass of the score trunk in the CHAIN form (trunk_chain.h) for large launches: what score_bwd.h does for
// one 16-row tile through LDS, one WAVE does here for its 16 * PT rows in registers.
//
// The reference gets these quantities from autograd - the score of the ENERGY model is the gradient of its inner-product energy
// (networks/gf_algorithms/energynet.py:200-222), the likelihood ODE needs the Skilling-Hutchinson divergence estimate
// (samplers.py:49-71) - i.e. a second network pass per evaluation.  Here:
//   * the forward pass is trunk_chain.h's (same ring, same slices, same instruction slots) and additionally records, one BIT per
//     element, which hidden / head activations are positive (h1, h2: 64 bits per 16-row chunk each, the three heads 192);
//   * backward through the stacked head layer: the seed of head chunk c,  g3[c] = [a3[c] > 0] * sum_i w_out[i][c] * u_i  (u = the
//     three output-layer cotangents of the head; w_out rows from the LDS table the forward epilogue uses), has the D-fragment
//     layout, i.e. it IS the B operand of k-group c of  g2 = Wx^T g3:  48 ring steps, one k-group x all 16 output chunks each
//     (the transposed pack w_headx_t is k-group major: a step's 16 fragments are one contiguous 16 KB slice), accumulating into
//     16 chunk accumulators; the seed of chunk c + 1 is computed in the instruction slots of step c;
//   * g2 . [h2 > 0] is again a B operand: pose_encoder.2^T as two half-layers of eight steps (the forward's slicing of a 256 x 256
//     layer), then g1 . [h1 > 0] through pose_encoder.0^T: ONE slice (16 k-groups x 1 output chunk), four partial accumulators.
// 130 ring slices per evaluation (65 forward + 48 + 16 + 1), ~2 x the forward's MFMAs, no activation or gradient ever in LDS.
// The pre-activations are the tile form's bit for bit (same MFMA order); gx sums its 256 products in a different order (four partial
// sums) than the tile form's single chain: 1e-7 relative.
#pragma once
#include "trunk_chain.h"

namespace gp_chain {

constexpr int NS_HEAD_T = HEADS / 16, NS_POSE2_T = 16, NSLICES_VJP = NSLICES + NS_HEAD_T + NS_POSE2_T + 1;

// LDS behind the forward's (Cfg<PT>): two per-THREAD tables, [entry][thread] (lane-linear: conflict free, private to the thread that
// wrote them - no barrier involved), for values that are written once in the forward pass and read with a RUN-TIME index in the
// backward pass (a register array would have to be indexed dynamically):
//   U  [9][PT]   the cotangent u of the row's nine outputs (the caller stores it: store_cotangent)
//   M3 [7][PT]   sign bits of the head activations, one 32-bit word per (head, half-layer) = 8 chunks x 4 elements; entry 6 = dump
template <int PT>
struct CfgV {
    static constexpr int OFF_U = Cfg<PT>::TOTAL, OFF_M3 = OFF_U + POSE * PT * NT, TOTAL = OFF_M3 + 7 * PT * NT;
    static constexpr size_t LDS_BYTES = (size_t)TOTAL * sizeof(float);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};
template <int PT>
__device__ __forceinline__ void store_cotangent(float *lds, const float (&u)[PT][POSE]) {
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int j = 0; j < POSE; ++j) lds[CfgV<PT>::OFF_U + (j * PT + p) * NT + threadIdx.x] = u[p][j];
}

// slice s of the forward + backward weight stream
__device__ __forceinline__ SliceSrc slice_src_vjp(const gp_scorenet &net, int s) {
    s = s < NSLICES_VJP ? s : NSLICES_VJP - 1;  // the ring runs ahead of the last step: those requests re-read the last slice (never used)
    if (s < NSLICES) return slice_src(net, s);
    SliceSrc r;
    if (s < NSLICES + NS_HEAD_T) {  // head layer transposed: k-group = head chunk s - NSLICES, all 16 output chunks
        r.b0 = reinterpret_cast<const f32x4 *>(net.w_headx_t) + (size_t)(s - NSLICES) * 16 * 64;
        r.b1 = r.b0 + 8 * 64;
        return r;
    }
    if (s < NSLICES + NS_HEAD_T + NS_POSE2_T) {  // pose_encoder.2 transposed: two k-groups x eight output chunks, as the forward layer
        const int q = s - NSLICES - NS_HEAD_T, c0 = 8 * (q >> 3), j = q & 7;
        const f32x4 *base = reinterpret_cast<const f32x4 *>(net.w_pose2_t);
        r.b0 = base + ((size_t)(2 * j) * (HID / 16) + c0) * 64;
        r.b1 = base + ((size_t)(2 * j + 1) * (HID / 16) + c0) * 64;
        return r;
    }
    r.b0 = reinterpret_cast<const f32x4 *>(net.w_pose0_t);  // pose_encoder.0 transposed: 16 k-groups x its one (zero-padded) output chunk
    r.b1 = r.b0 + 8 * 64;
    return r;
}

// ring_step of trunk_chain.h on the forward + backward stream (the slice requested three steps ahead comes from slice_src_vjp).
// K16 = false: NB sub-blocks x (16 / NB) output chunks, acc[p][chunk].  K16 = true (the last step): the 16 fragments are the 16
// k-groups of ONE output chunk - fragment 4 q + u multiplies hk[4 q + u] into partial accumulator acc[p][u].
template <int PT, int NB, bool K16, class Side>
__device__ __forceinline__ void ring_step_v(State<PT> &st, f32x4 *ring, const gp_scorenet &net, int s, const f32x4 (&hk)[NB][PT],
                                            f32x4 (&acc)[PT][K16 ? 4 : 16 / NB], Side side) {
    using C = Cfg<PT>;
    static_assert(!K16 || NB == 16, "the k-group step takes all 16 k-groups");
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int NCH = K16 ? 1 : 16 / NB;
    const f32x4 *slot = ring + (s % C::NR) * SLICE, *nslot = ring + ((s + 1) % C::NR) * SLICE;
    f32x4 *dst = ring + ((s + C::W) % C::NR) * SLICE;
    const SliceSrc src = slice_src_vjp(net, s + C::W + 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = (4 * q) / NCH, n0 = K16 ? 0 : (4 * q) % NCH;
        f32x4 wf[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wf[u] = st.wpre[u];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int k = 4 * q + jj;
            st.wpre[jj] = q < 3 ? slot[(4 * (q + 1) + jj) * 64 + lane] : nslot[jj * 64 + lane];
            if (k < 8 && (k & 1)) dst[tid + (k >> 1) * NT] = st.hold[k >> 1];
            if (k >= 8 && (k & 1)) st.hold[(k - 8) >> 1] = slice_elem(src, tid + ((k - 8) >> 1) * NT);
            side(k);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    if constexpr (K16)
                        acc[p][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][jj], hk[b + u][p][jj], acc[p][u], 0, 0, 0);
                    else
                        acc[p][n0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][jj], hk[b][p][jj], acc[p][n0 + u], 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < 4 * PT + 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x096, OTHER_PER_MFMA, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (s % C::BP == C::BP - 1) ring_barrier();
}

// four sign bits of a fragment at bit position `sh` (a multiple of 4) of a mask word
__device__ __forceinline__ uint32_t pos_bits(const f32x4 &v, int sh) {
    const uint32_t b = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
    return b << sh;
}
__device__ __forceinline__ f32x4 keep_where(const f32x4 &v, uint32_t bits4) {
    return f32x4{(bits4 & 1u) ? v.x : 0.f, (bits4 & 2u) ? v.y : 0.f, (bits4 & 4u) ? v.z : 0.f, (bits4 & 8u) ? v.w : 0.f};
}

// f_theta (+ output bias) AND gx = J_f^T u of this wave's rows.
//   xf[p]      the row's pose as the B fragment of k-group 0 (lane (row, g): components 4g .. 4g+3, zero beyond 8)
//   the cotangent u of the nine outputs of the row (the same in every lane of the row): in the LDS table (store_cotangent) beforehand
//   f[p][0..8] valid in every lane;  gx[p]: lane (row, g) holds components 4g .. 4g+3 of J_f^T u (zero beyond 8)
// All four waves of the workgroup call this together; begin_request / begin_deposit of trunk_chain.h start the ring (its first
// slices are the forward's); the kernel's dynamic LDS is CfgV<PT>::LDS_BYTES.
template <int PT>
__device__ __forceinline__ void run_vjp(State<PT> &st, float *lds, const gp_scorenet &net, const f32x4 (&xf)[PT], float (&f)[PT][POSE], f32x4 (&gx)[PT]) {
    using C = Cfg<PT>;
    using V = CfgV<PT>;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    f32x4 *ring = reinterpret_cast<f32x4 *>(lds);
    const float *ul = lds + V::OFF_U + threadIdx.x;
    uint32_t *m3l = reinterpret_cast<uint32_t *>(lds) + V::OFF_M3 + threadIdx.x;
    uint32_t m1[PT][2], m2[PT][2], mcur[PT];  // positive-activation bits: chunk n, element e of a 256-wide layer at bit 4 n + e
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        m1[p][0] = m1[p][1] = m2[p][0] = m2[p][1] = mcur[p] = 0u;
#pragma unroll
        for (int j = 0; j < POSE; ++j) f[p][j] = 0.f;
    }
    __syncthreads();  // prologue LDS writes (first slots, staged operands) are visible
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) st.wpre[uu] = ring[uu * 64 + lane];
    f32x4 h2[PT][16];
    {
        f32x4 h1[PT][16];
        // ---- step 0: pose_encoder.0
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int n = 0; n < 16; ++n) h1[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            f32x4 hk[1][PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) hk[0][p] = xf[p];
            ring_step_v<PT, 1, false>(st, ring, net, 0, hk, h1, NoSide());
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + C::OFF_B0 + 16 * n + 4 * g);
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                h1[p][n] = relu4(h1[p][n] + bv);
                m1[p][n >> 3] |= pos_bits(h1[p][n], 4 * (n & 7));
                asm volatile("" : "+v"(m1[p][n >> 3]));  // computed HERE (the optimiser otherwise keeps the activations alive - spilled - and forms the bits at their use)
            }
        }
        // ---- steps 1..16: pose_encoder.2 as two half-layers
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 acc[PT][8];
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int n = 0; n < 8; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 hk[2][PT];
#pragma unroll
                for (int p = 0; p < PT; ++p) hk[0][p] = h1[p][2 * j], hk[1][p] = h1[p][2 * j + 1];
                ring_step_v<PT, 2, false>(st, ring, net, 1 + 8 * half + j, hk, acc, NoSide());
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + C::OFF_B2 + 16 * (8 * half + n) + 4 * g);
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    h2[p][8 * half + n] = relu4(acc[p][n] + bv);
                    m2[p][half] |= pos_bits(h2[p][8 * half + n], 4 * n);
                    asm volatile("" : "+v"(m2[p][half]));
                }
            }
        }
    }
    // ---- steps 17..64: the three heads, exactly as trunk_chain.h (ping-pong accumulator sets, output layers as v_mfma_f32_4x4x1 in
    // the slots of the next half-layer); the epilogue additionally records the sign bits of the head activations
    f32x4 oacc[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) oacc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    HeadEpi<PT> e;
    const int m = lane & 3;
    auto epi_slot = [&](int k, const f32x4 (&done)[PT][8], int n, int hd, int halfd, bool real) {
        const int col = 128 * halfd + 16 * n + 4 * g;
        if (k == 0) e.w0 = *reinterpret_cast<const f32x4 *>(lds + C::OFF_WOUT + (m < 3 ? 3 * hd + m : POSE) * HID + col);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            if (k == 3 + p) e.cv[p] = *reinterpret_cast<const f32x4 *>(lds + C::OFF_CVT + (st.cloud[p] - st.cloud0) * HEADS + 256 * hd + col);
            if (k == 5 + p) {
                e.v[p] = relu4(done[p][n] + e.cv[p]);
                asm volatile("" : "+v"(e.v[p]));
                // the eight chunks of a half-layer fill exactly one word; the finished word goes to the thread's LDS table (the zero
                // accumulators consumed before the first head: to the dump entry)
                mcur[p] |= pos_bits(e.v[p], 4 * n);
                asm volatile("" : "+v"(mcur[p]));
                if (n == 7) {
                    m3l[((real ? 2 * hd + halfd : 6) * PT + p) * NT] = mcur[p];
                    mcur[p] = 0u;
                }
            }
            if (k == 7 + p) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) oacc[p] = __builtin_amdgcn_mfma_f32_4x4x1f32(e.w0[jj], e.v[p][jj], oacc[p], 0, 0, 0);
            }
        }
    };
    auto finish_head = [&](int hd) {
#pragma unroll
        for (int p = 0; p < PT; ++p) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = oacc[p][c];
                f[p][c] = hd == 0 ? v : f[p][c];
                f[p][3 + c] = hd == 1 ? v : f[p][3 + c];
                f[p][6 + c] = hd == 2 ? v : f[p][6 + c];
            }
            oacc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 accB[PT][8];  // outlives the head loop: its last contents (head 2, second half) are consumed under the first backward steps
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int n = 0; n < 8; ++n) accB[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        f32x4 accA[PT][8];
#pragma unroll 1
        for (int h = 0; h < 3; ++h) {
            const int hprev = h > 0 ? h - 1 : 0;
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int n = 0; n < 8; ++n) accA[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 hk[2][PT];
#pragma unroll
                for (int p = 0; p < PT; ++p) hk[0][p] = h2[p][2 * j], hk[1][p] = h2[p][2 * j + 1];
                ring_step_v<PT, 2, false>(st, ring, net, 17 + 16 * h + j, hk, accA, [&](int k) { epi_slot(k, accB, j, hprev, 1, h > 0); });
            }
            finish_head(h - 1);
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int n = 0; n < 8; ++n) accB[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 hk[2][PT];
#pragma unroll
                for (int p = 0; p < PT; ++p) hk[0][p] = h2[p][2 * j], hk[1][p] = h2[p][2 * j + 1];
                ring_step_v<PT, 2, false>(st, ring, net, 25 + 16 * h + j, hk, accB, [&](int k) { epi_slot(k, accA, j, h, 0, true); });
            }
        }
    }
    // ---- backward through the head layer: steps 65 .. 112, head chunk c = k-group c of g2 = Wx^T g3.  The seed of chunk c + 1 is
    // computed in the slots of step c; the forward's last half-layer epilogue (head 2, second half, still in accB) rides in the slots
    // of the first eight backward steps instead of running exposed behind the last forward step.
    f32x4 g2a[PT][16];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int n = 0; n < 16; ++n) g2a[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 w0, w1, w2, g3n[PT], g3c[PT];
    float us[PT][3];
    uint32_t mw[PT];
    // seed of head chunk c (head hh = c / 16, columns 16 (c % 16) + 4 g .. of the head) for the lane's rows.  Chunk c of heads 0 / 1
    // reads bits recorded long ago; those of head 2's second half are recorded by the epilogue riding in steps 65..72, eight steps and
    // more ahead of their seeds (chunks 40..47).
    auto seed_slot = [&](int k, int c) {
        const int hh = c >> 4, cc = c & 15;
        if (k == 1) {
            const float *wo = lds + C::OFF_WOUT + 3 * hh * HID + 16 * cc + 4 * g;
            w0 = *reinterpret_cast<const f32x4 *>(wo);
            w1 = *reinterpret_cast<const f32x4 *>(wo + HID);
            w2 = *reinterpret_cast<const f32x4 *>(wo + 2 * HID);
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            if (k == 3 + 2 * p) {
                us[p][0] = ul[((3 * hh + 0) * PT + p) * NT];
                us[p][1] = ul[((3 * hh + 1) * PT + p) * NT];
                us[p][2] = ul[((3 * hh + 2) * PT + p) * NT];
                mw[p] = m3l[((2 * hh + (cc >> 3)) * PT + p) * NT];
            }
            if (k == 9 + 2 * p) {
                const float u0 = us[p][0], u1 = us[p][1], u2 = us[p][2];
                const uint32_t bits = mw[p] >> (4 * (cc & 7));
                // the tile form's expression (score_bwd.h): (w0 u0 + w1 u1) + w2 u2
                const f32x4 sv = f32x4{(w0.x * u0 + w1.x * u1) + w2.x * u2, (w0.y * u0 + w1.y * u1) + w2.y * u2,
                                       (w0.z * u0 + w1.z * u1) + w2.z * u2, (w0.w * u0 + w1.w * u1) + w2.w * u2};
                g3n[p] = keep_where(sv, bits);
                asm volatile("" : "+v"(g3n[p]));
            }
        }
    };
#pragma unroll
    for (int k = 0; k < 16; ++k) seed_slot(k, 0);  // chunk 0: outside the stream
#pragma unroll
    for (int p = 0; p < PT; ++p) g3c[p] = g3n[p];
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // static j: the forward epilogue's chunk index
        f32x4 hk[1][PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) hk[0][p] = g3c[p];
        ring_step_v<PT, 1, false>(st, ring, net, NSLICES + j, hk, g2a, [&](int k) {
            epi_slot(k, accB, j, 2, 1, true);
            seed_slot(k, j + 1);
        });
#pragma unroll
        for (int p = 0; p < PT; ++p) g3c[p] = g3n[p];
    }
    finish_head(2);
    // the four lane groups each hold the sum over their quarter of the channels
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int j = 0; j < POSE; ++j) {
            float v = f[p][j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            f[p][j] = v + net.b_out[j];
        }
#pragma unroll 1
    for (int c = 8; c < NS_HEAD_T; ++c) {
        f32x4 hk[1][PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) hk[0][p] = g3c[p];
        const int cn = c + 1 < NS_HEAD_T ? c + 1 : NS_HEAD_T - 1;
        ring_step_v<PT, 1, false>(st, ring, net, NSLICES + c, hk, g2a, [&](int k) { seed_slot(k, cn); });
#pragma unroll
        for (int p = 0; p < PT; ++p) g3c[p] = g3n[p];
    }
    // ---- g2 . [h2 > 0], then pose_encoder.2 transposed (steps 113 .. 128, two half-layers) -> g1 . [h1 > 0]
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int n = 0; n < 16; ++n) g2a[p][n] = keep_where(g2a[p][n], m2[p][n >> 3] >> (4 * (n & 7)));
    f32x4 g1k[16][PT];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 acc[PT][8];
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 hk[2][PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) hk[0][p] = g2a[p][2 * j], hk[1][p] = g2a[p][2 * j + 1];
            ring_step_v<PT, 2, false>(st, ring, net, NSLICES + NS_HEAD_T + 8 * half + j, hk, acc, NoSide());
        }
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int p = 0; p < PT; ++p) g1k[8 * half + n][p] = keep_where(acc[p][n], m1[p][half] >> (4 * n));
    }
    // ---- pose_encoder.0 transposed: one slice = the 16 k-groups of the single output chunk, four partial accumulators
    f32x4 acc4[PT][4];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc4[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    ring_step_v<PT, 16, true>(st, ring, net, NSLICES_VJP - 1, g1k, acc4, NoSide());
#pragma unroll
    for (int p = 0; p < PT; ++p) gx[p] = (acc4[p][0] + acc4[p][1]) + (acc4[p][2] + acc4[p][3]);
}

}  // namespace gp_chain
