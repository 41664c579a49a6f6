// Register-resident form of the PoseScoreNet / PoseEnergyNet trunk (scorenet.py:178-222) for LARGE launches.
//
// The tile form (score_trunk.h) gives a 16- or 32-row tile to a whole workgroup: every layer round-trips its activations through
// LDS behind a barrier, every workgroup streams the full 1 MB weight set L2 -> VGPR, and the per-row work (sampler update,
// output combine) runs on 16 or 32 of its 256 / 512 threads.  Here ONE WAVE owns 16 * PT rows from the sampler update to the
// score, and the activations never leave its registers:
//   * for v_mfma_f32_16x16x4_f32 with the weights as the A operand, the D fragment (lane = row, four consecutive channels of
//     16-channel chunk nc) IS the B fragment the next layer needs for k-group nc, so 9 -> 256 -> 256 -> 768 chains in registers
//     with no LDS activation traffic and no barrier between the layers of a row;
//   * the weights stream through a SIX-slot LDS ring shared by the four waves of the workgroup (one per SIMD, the whole 512-entry
//     register file each), 65 slices of 16 KB (16 fragments of 64 lanes x 16 B) - one pass over the weights serves 64 * PT rows
//     instead of 32; the slice three steps ahead is written while the current one is multiplied, ONE bare s_barrier every second
//     step (geometry and its safety conditions: Cfg below);
//   * the three Linear(256, 3) output layers run on the matrix pipe as well, as v_mfma_f32_4x4x1_16b_f32 (sixteen independent 4 x 4
//     outer products, 8 cycles each): block b = lanes 4b .. 4b+3 = four rows of ONE lane group; A = the head's three output rows
//     (+ a row of zeros) at the channel this lane group holds, B = the post-ReLU head activations exactly as the accumulators hold
//     them; one output chunk per ring step of the NEXT half-layer (two accumulator sets, ping-pong), so the epilogue runs in the
//     shadow of the MFMAs.  Lane (row, g) collects its row's three outputs over the channels of lane group g; the four lane groups
//     are summed once per launch (two __shfl_xor steps on nine values).
// A ring step is a hand-placed instruction stream: 16 slots of 4 * PT MFMAs with a little other work each (one weight fragment
// request, a quarter of the ring refill, a piece of the previous half-layer's epilogue), pinned to its slot with scheduling
// barriers and interleaved with the slot's MFMAs by sched_group_barrier - with one wave per SIMD nothing else hides a burst of
// non-matrix instructions (measured: the same work issued as one block per step cost 9 % of the launch, the epilogues, which the
// optimiser sinks to the end of a half-layer unless they are pinned, another 6 %).
// The MFMA sequence per hidden accumulator (k-group major, jj = 0..3) is the tile form's, so pre-activations agree bit for bit; only
// the order in which the 256 products of an output component are summed differs (1e-7 relative).
// Measured (MI355X, 32 000 rows): 137-139 us per launch (0.78-0.79 of the fp32 MFMA peak) against 146.6 us for the 32-row tile form;
// ring steps run at 0.90-0.92 of the MFMA issue rate (what a bare v_mfma loop with LDS operand reads reaches with one wave per SIMD,
// scratch/occ/mfma_power.hip), the rest is the prologue (operand requests, ring start-up: 4 %), 250 workgroups on 256 CUs, the
// sustained clock and the tail (history in EXPERIMENTS.md §G).
#pragma once
#include "score_trunk.h"

namespace gp_chain {

using namespace gp_trunk;

constexpr int SLICE = 16 * 64;        // f32x4 per ring slice (16 KB) = 16 weight fragments of 64 lanes
constexpr int NSLICES = 1 + 16 + 48;  // pose_encoder.0 | pose_encoder.2 | three heads
constexpr int NCL = 4;                // clouds whose (cvec + tvec) rows are staged in LDS: a workgroup's rows must span <= NCL clouds (fits())
constexpr int NW = 4, NT = 64 * NW, PER_T = SLICE / NT;  // four waves, one per SIMD; f32x4 each thread moves per slice
constexpr int OTHER_PER_MFMA = 3;     // other instructions the scheduler may place behind each MFMA of a slot (2 and 4 measured equal)

// Ring geometry: NR slices of 16 KB; slice t sits at position t % NR.  In step s the slice requested one step earlier (s + W) is
// written, slice s + W + 1 is requested, and the first fragments of slice s + 1 are read at the end; a barrier closes every BP-th
// step.  Safe when W >= BP + 1 (the slice whose first fragments are read at the end of step s was written at step s + 1 - W: a
// barrier lies between) and NR >= W + BP (position (s + W) % NR was last read in step s + W - NR: a barrier lies between).
// Shipped: PT = 2 (32 rows per wave, 128 per workgroup, one workgroup per CU with the whole register file), NR 6, W 3, BP 2.
// (Measured and dropped: PT = 1 with NR 3 / W 2 / BP 1 fits two workgroups per CU - 71 KB, 244 registers - but a co-resident pair
// takes 160 us against 2 x 84 us one after the other: two waves per SIMD contend for the matrix pipe instead of filling each
// other's bubbles.  One barrier per step instead of every second: equal.)
template <int PT>
struct Cfg {
    static constexpr int ROWS = 16 * NW * PT;
    static constexpr int NR = 6, W = 3, BP = 2;
    static_assert(W >= BP + 1 && NR >= W + BP, "ring geometry");
    // k candidates per cloud: the rows of a workgroup span at most ceil((ROWS - 1) / k) + 1 clouds
    static constexpr bool fits(int k) { return k > 0 && (ROWS - 2 + k) / k + 1 <= NCL; }
    // LDS (floats): ring [NR][SLICE][4] | w_out [10][256] (row 9 = zeros) | b_pose0 [256] | b_pose2 [256] | cvt [NCL][768] = cvec[cloud] + tvec
    static constexpr int OFF_WOUT = NR * SLICE * 4, OFF_B0 = OFF_WOUT + (POSE + 1) * HID, OFF_B2 = OFF_B0 + HID, OFF_CVT = OFF_B2 + HID,
                         TOTAL = OFF_CVT + NCL * HEADS;
    static constexpr size_t LDS_BYTES = (size_t)TOTAL * sizeof(float);
};

// Slice s of the weight stream.  Slice 0 = pose_encoder.0: its one (zero-padded) k-group, 16 output chunks.  Every other slice =
// TWO k-groups x EIGHT output chunks (two runs of 8 KB in the packed layout [k-group][chunk][lane][4]), so that a ring step
// accumulates into 8 chunk accumulators (32 registers per 16 rows) and a 256-wide layer is two half-layers of 8 steps:
//   s = 1 + 8 half + j            pose_encoder.2, output chunks [8 half, +8), k-groups 2j, 2j+1
//   s = 17 + 16 h + 8 half + j    head h (columns [256 h, +256) of the stacked first head layer), the same split
// Sub-block b (k-group 2j + b) occupies fragments [8 b, 8 b + 8) of the slice.
struct SliceSrc {
    const f32x4 *b0, *b1;  // the two runs of 8 chunks x 64 lanes
};
__device__ __forceinline__ SliceSrc slice_src(const gp_scorenet &net, int s) {
    s = s < NSLICES ? s : NSLICES - 1;  // the ring runs two slices ahead: requests past the end re-read the last slice (never used)
    SliceSrc r;
    if (s == 0) {
        r.b0 = reinterpret_cast<const f32x4 *>(net.w_pose0);
        r.b1 = r.b0 + 8 * 64;
        return r;
    }
    const bool l2 = s <= 16;
    const int q = l2 ? s - 1 : s - 17;
    const f32x4 *base = reinterpret_cast<const f32x4 *>(l2 ? net.w_pose2 : net.w_headx);
    const int nc = l2 ? HID / 16 : HEADS / 16, c0 = 8 * (q >> 3), j = q & 7;  // heads: 8 (2 h + half) = 16 h + 8 half
    r.b0 = base + ((size_t)(2 * j) * nc + c0) * 64;
    r.b1 = base + ((size_t)(2 * j + 1) * nc + c0) * 64;
    return r;
}
// element e (0 .. SLICE-1) of a slice: fragments [0, 8) from run b0, [8, 16) from run b1
__device__ __forceinline__ f32x4 slice_elem(const SliceSrc &src, int e) { return e < SLICE / 2 ? src.b0[e] : src.b1[e - SLICE / 2]; }

template <int PT>
struct State {
    f32x4 hold[PER_T];  // slice in flight to the ring (global -> registers -> LDS)
    f32x4 wpre[4];      // first fragment group of the upcoming ring step
    int cloud[PT];      // cloud of this lane's row in p-chunk p
    int cloud0;         // first cloud of the workgroup's rows
};

// Ring prologue in two halves, so that the caller's own work runs between the request and the first use:
//   begin_request()  asks for the first W slices, slice W (-> hold) and the staged epilogue operands - w_out (+ a row of zeros: the
//                    output layers' A operand has four rows of which three are real), the two hidden biases, cvec of every cloud the
//                    workgroup's rows touch and tvec (requires Cfg<PT>::fits(kcand)).  Issue the per-row operand requests BEFORE it:
//                    memory returns in order, so whatever the caller needs first must be asked for first;
//   begin_deposit()  writes them to LDS; run() starts with the barrier that publishes these writes.
//   wg_row0: first row of the workgroup; row_end: one past the last valid row it may touch (rows beyond are clamped duplicates)
constexpr int N_WOUT = ((POSE + 1) * HID / 4 + NT - 1) / NT, N_CVT = NCL * (HEADS / 4) / NT;
static_assert(NCL * (HEADS / 4) % NT == 0 && HID / 4 <= NT, "staging loops");
template <int PT>
struct Staged {
    f32x4 first[Cfg<PT>::W][PER_T];
    f32x4 wout[N_WOUT], b0, b2, cv[N_CVT], tv[N_CVT];
};
template <int PT>
__device__ __forceinline__ void begin_request(State<PT> &st, Staged<PT> &sg, const gp_scorenet &net, const float *__restrict__ cvec,
                                              const float *__restrict__ tvec, int wg_row0, int row_end, int kcand) {
    using C = Cfg<PT>;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int t = 0; t < C::W; ++t) {
        const SliceSrc src = slice_src(net, t);
#pragma unroll
        for (int u = 0; u < PER_T; ++u) sg.first[t][u] = slice_elem(src, tid + u * NT);
    }
    {
        const SliceSrc src = slice_src(net, C::W);
#pragma unroll
        for (int u = 0; u < PER_T; ++u) st.hold[u] = slice_elem(src, tid + u * NT);
    }
    st.cloud0 = wg_row0 / kcand;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        int r = wg_row0 + (wave * PT + p) * 16 + (lane & 15);
        r = r < row_end ? r : row_end - 1;
        st.cloud[p] = r / kcand;
    }
#pragma unroll
    for (int u = 0; u < N_WOUT; ++u) {
        const int f = tid + u * NT;
        sg.wout[u] = f < POSE * HID / 4 ? reinterpret_cast<const f32x4 *>(net.w_out)[f] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < HID / 4) {
        sg.b0 = reinterpret_cast<const f32x4 *>(net.b_pose0)[tid];
        sg.b2 = reinterpret_cast<const f32x4 *>(net.b_pose2)[tid];
    }
    const int last_cloud = (row_end - 1) / kcand;
#pragma unroll
    for (int u = 0; u < N_CVT; ++u) {
        const int f = tid + u * NT, c = f / (HEADS / 4), o = f - c * (HEADS / 4);
        int cl = st.cloud0 + c;
        cl = cl < last_cloud ? cl : last_cloud;
        sg.cv[u] = reinterpret_cast<const f32x4 *>(cvec + (size_t)cl * HEADS)[o];
        sg.tv[u] = reinterpret_cast<const f32x4 *>(tvec)[o];
    }
}
template <int PT>
__device__ __forceinline__ void begin_deposit(const Staged<PT> &sg, float *lds) {
    using C = Cfg<PT>;
    const int tid = threadIdx.x;
    f32x4 *ring = reinterpret_cast<f32x4 *>(lds);
#pragma unroll
    for (int u = 0; u < N_WOUT; ++u) {
        const int f = tid + u * NT;
        if (f < (POSE + 1) * HID / 4) reinterpret_cast<f32x4 *>(lds + C::OFF_WOUT)[f] = sg.wout[u];
    }
    if (tid < HID / 4) {
        reinterpret_cast<f32x4 *>(lds + C::OFF_B0)[tid] = sg.b0;
        reinterpret_cast<f32x4 *>(lds + C::OFF_B2)[tid] = sg.b2;
    }
#pragma unroll
    for (int u = 0; u < N_CVT; ++u) reinterpret_cast<f32x4 *>(lds + C::OFF_CVT)[tid + u * NT] = sg.cv[u] + sg.tv[u];
#pragma unroll
    for (int t = 0; t < C::W; ++t)
#pragma unroll
        for (int u = 0; u < PER_T; ++u) ring[t * SLICE + tid + u * NT] = sg.first[t][u];
}
template <int PT>
__device__ __forceinline__ void begin(State<PT> &st, float *lds, const gp_scorenet &net, const float *__restrict__ cvec, const float *__restrict__ tvec,
                                      int wg_row0, int row_end, int kcand) {
    Staged<PT> sg;
    begin_request<PT>(st, sg, net, cvec, tvec, wg_row0, row_end, kcand);
    begin_deposit<PT>(sg, lds);
}

// End of a ring step.  What the ring needs from this barrier: (i) every wave's reads of the slot it just multiplied have
// RETURNED before any wave overwrites it in the next step - they have, the MFMAs consumed them; (ii) the slice written during
// this step is visible two steps later - LDS operations of a wave complete in order, and younger reads have returned.  The first
// fragment group of the next step, requested in the last slots of this one, reads a slot nobody writes before the NEXT barrier and
// may stay in flight: no s_waitcnt lgkmcnt(0) here (the fence of __syncthreads() costs one exposed LDS round trip per step).
// The empty asm statements keep the compiler from moving LDS accesses across the barrier.
__device__ __forceinline__ void ring_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }
__device__ __forceinline__ float dot4(const f32x4 &v, const f32x4 &w) { return v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w; }

// operands of a head epilogue in flight between the slots of a ring step
template <int PT>
struct HeadEpi {
    f32x4 w0, cv[PT], v[PT];
};

// One ring step: acc[p][n] += W[n][k-group b] . hk[b][p] over a slice of NB sub-blocks x (16 / NB) output chunks, as 16 slots of
// 4 * PT MFMAs (slot k = 4 q + jj: fragment group q, k-step jj).  Before the MFMAs of slot k:
//   * fragment jj of group q + 1 (or of the NEXT step's first group, slot `nslot`, published by the previous barrier) is requested;
//   * slots 1, 3, 5, 7: a quarter of the slice held in registers (s + 2) goes to its ring slot - last read in step s - 1, which
//     every wave has left; slots 9, 11, 13, 15: a quarter of slice s + 3 is requested (in flight for a whole step);
//   * `side(k)`: the caller's piece of epilogue work for this slot.
// The scheduling barriers pin every piece where it is written.
template <int PT, int NB, class Side>
__device__ __forceinline__ void ring_step(State<PT> &st, f32x4 *ring, const gp_scorenet &net, int s, const f32x4 (&hk)[NB][PT],
                                          f32x4 (&acc)[PT][16 / NB], Side side) {
    using C = Cfg<PT>;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int NCH = 16 / NB;  // output chunks of the step
    const f32x4 *slot = ring + (s % C::NR) * SLICE, *nslot = ring + ((s + 1) % C::NR) * SLICE;
    f32x4 *dst = ring + ((s + C::W) % C::NR) * SLICE;
    const SliceSrc src = slice_src(net, s + C::W + 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // fragment group [4 q, 4 q + 4) of the slice: sub-block 4 q / NCH, chunks (4 q % NCH) ..
        const int b = (4 * q) / NCH, n0 = (4 * q) % NCH;
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
                for (int p = 0; p < PT; ++p) acc[p][n0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][jj], hk[b][p][jj], acc[p][n0 + u], 0, 0, 0);
            // the slot's other instructions go BETWEEN its MFMAs, at most three behind each (an MFMA occupies the matrix pipe for 32
            // cycles; a burst of more than ~6 other issues behind it leaves the pipe idle - with one wave per SIMD nothing fills it)
#pragma unroll
            for (int i = 0; i < 4 * PT + 4; ++i) {  // + 4: a slot of the head epilogue carries four MFMAs of its own
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x096, OTHER_PER_MFMA, 0);  // VALU | SALU | VMEM | DS
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (s % C::BP == C::BP - 1) ring_barrier();
}

struct NoSide {
    __device__ __forceinline__ void operator()(int) const {}
};

// f_theta (+ output bias) of this wave's rows.  xf[p] = the row's pose as the B fragment of k-group 0: lane (row, g) holds
// components 4g .. 4g+3 (zero beyond 8).  f[p][0..8] is valid in the lanes of lane group 0 (lane = row).
// All four waves of the workgroup must call this together (NSLICES + 1 barriers).
template <int PT>
__device__ __forceinline__ void run(State<PT> &st, float *lds, const gp_scorenet &net, const f32x4 (&xf)[PT], float (&f)[PT][POSE]) {
    using C = Cfg<PT>;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    f32x4 *ring = reinterpret_cast<f32x4 *>(lds);
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int j = 0; j < POSE; ++j) f[p][j] = 0.f;
    __syncthreads();  // prologue LDS writes (slots 0 and 1, staged operands) are visible
#pragma unroll
    for (int u = 0; u < 4; ++u) st.wpre[u] = ring[u * 64 + lane];
    f32x4 h1[PT][16], h2[PT][16];
    // ---- step 0: pose_encoder.0 (9 -> 256, one zero-padded k-group, all 16 output chunks)
    {
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int n = 0; n < 16; ++n) h1[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 hk[1][PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) hk[0][p] = xf[p];
        ring_step<PT, 1>(st, ring, net, 0, hk, h1, NoSide());
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + C::OFF_B0 + 16 * n + 4 * g);
#pragma unroll
            for (int p = 0; p < PT; ++p) h1[p][n] = relu4(h1[p][n] + bv);
        }
    }
    // ---- steps 1..16: pose_encoder.2 (256 -> 256) as two half-layers of 8 output chunks
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
            ring_step<PT, 2>(st, ring, net, 1 + 8 * half + j, hk, acc, NoSide());
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + C::OFF_B2 + 16 * (8 * half + n) + 4 * g);
#pragma unroll
            for (int p = 0; p < PT; ++p) h2[p][8 * half + n] = relu4(acc[p][n] + bv);
        }
    }
    // ---- steps 17..64: the three heads (256 -> 256 each) as six half-layers.  Two accumulator sets: while half-layer i accumulates
    // into one, the Linear(256, 3) epilogue of half-layer i - 1 consumes the other, one output chunk per ring step.
    // The Linear(256, 3) output layer of a head runs on the matrix pipe too, as v_mfma_f32_4x4x1_16b_f32 (16 independent 4 x 4 outer
    // products, 8 cycles): block b = lanes 4b .. 4b+3 = four rows of ONE lane group; A = the head's three rows (+ the zero row of the
    // LDS table for lane & 3 == 3) at the channel this lane group holds, B = the post-ReLU head activations exactly as the accumulators
    // hold them.  Lane (row, g) collects the three outputs of its row over the channels of lane group g; the four groups are summed
    // once at the very end.  (As a zero-padded 16-row v_mfma_f32_16x16x4_f32 operand the same products cost four times the pipe time.)
    f32x4 oacc[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) oacc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    HeadEpi<PT> e;
    const int m = lane & 3;
    // epilogue piece of slot k for output chunk n of `done` (a finished half-layer: head hd, half `halfd`)
    auto epi_slot = [&](int k, const f32x4 (&done)[PT][8], int n, int hd, int halfd) {
        const int col = 128 * halfd + 16 * n + 4 * g;  // column within the head
        if (k == 0) e.w0 = *reinterpret_cast<const f32x4 *>(lds + C::OFF_WOUT + (m < 3 ? 3 * hd + m : POSE) * HID + col);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            if (k == 3 + p) e.cv[p] = *reinterpret_cast<const f32x4 *>(lds + C::OFF_CVT + (st.cloud[p] - st.cloud0) * HEADS + 256 * hd + col);
            // the empty asm statement pins the arithmetic to ITS slot (the optimiser otherwise sinks the whole epilogue to the end
            // of the half-layer, where nothing hides it)
            if (k == 5 + p) {
                e.v[p] = relu4(done[p][n] + e.cv[p]);
                asm volatile("" : "+v"(e.v[p]));
            }
            if (k == 7 + p) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) oacc[p] = __builtin_amdgcn_mfma_f32_4x4x1f32(e.w0[jj], e.v[p][jj], oacc[p], 0, 0, 0);
            }
        }
    };
    // this lane group's share of the output components of head hd
    auto finish_head = [&](int hd) {
#pragma unroll
        for (int p = 0; p < PT; ++p) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = oacc[p][c];
                // f is a register array: written with compile-time indices, the head selects
                f[p][c] = hd == 0 ? v : f[p][c];
                f[p][3 + c] = hd == 1 ? v : f[p][3 + c];
                f[p][6 + c] = hd == 2 ? v : f[p][6 + c];
            }
            oacc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 accA[PT][8], accB[PT][8];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int n = 0; n < 8; ++n) accB[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int h = 0; h < 3; ++h) {
        // half 0 of head h accumulates into accA; accB holds half 1 of head h - 1 (zeros before the first head: its "epilogue" adds
        // into partials that are discarded right after)
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
            ring_step<PT, 2>(st, ring, net, 17 + 16 * h + j, hk, accA, [&](int k) { epi_slot(k, accB, j, hprev, 1); });
        }
        finish_head(h - 1);  // h = 0: selects nothing, clears the partials
        // half 1 of head h accumulates into accB; accA (half 0 of head h) is consumed
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int n = 0; n < 8; ++n) accB[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 hk[2][PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) hk[0][p] = h2[p][2 * j], hk[1][p] = h2[p][2 * j + 1];
            ring_step<PT, 2>(st, ring, net, 25 + 16 * h + j, hk, accB, [&](int k) { epi_slot(k, accA, j, h, 0); });
        }
    }
    // tail: the last half-layer's epilogue has no MFMAs left to hide behind
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int k = 0; k < 16; ++k) epi_slot(k, accB, n, 2, 1);
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
}

// component j = 4 g + q of a per-lane 9-vector -> the B fragment of k-group 0 (zero beyond component 8)
__device__ __forceinline__ f32x4 pose_fragment(const float (&x)[POSE], int g) {
    f32x4 r;
    r.x = g == 0 ? x[0] : (g == 1 ? x[4] : (g == 2 ? x[8] : 0.f));
    r.y = g == 0 ? x[1] : (g == 1 ? x[5] : 0.f);
    r.z = g == 0 ? x[2] : (g == 1 ? x[6] : 0.f);
    r.w = g == 0 ? x[3] : (g == 1 ? x[7] : 0.f);
    return r;
}

}  // namespace gp_chain
