// Fused set-abstraction scale: neighbourhood gather -> 3-layer shared MLP (BN folded, ReLU) on fp32 MFMA ->
// max over the neighbourhood.  Replaces QueryAndGroup/GroupAll + SharedMLP + F.max_pool2d of the reference
// (pointnet2_utils.py:232-291, pytorch_utils.py:5-32, pointnet2_modules.py:37-52) and never materialises the
// grouped [B, C+3, np, ns] tensor (12 MB/cloud in the reference).
//
// One 256-thread workgroup owns P consecutive (centre, sample) rows of one cloud:
//   gather rows into LDS  [P][K0pad+8]   (feature columns first, then dx,dy,dz, zero pad)
//   layer 1: LDS A -> LDS B,  layer 2: LDS B -> LDS A,  layer 3: LDS A -> registers -> max -> global.
// Weights stream from L2 straight into MFMA A-operand registers (host-packed fragment order, gp_common.h).
#include "gp_common.h"

namespace {

struct SAArgs {
    int n, np, ns, cin, c1, c2, c3;
    const float *xyz, *feats_in, *new_xyz;
    const int32_t *idx;
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    float *out;
    int cout_total, cout_off, groupall;
};

// layer 3 + max over the neighbourhood, straight from the accumulators (WN waves along channels, PT p-chunks per wave)
// [nc_lo, nc_hi): the 16-channel output chunks this workgroup computes (all of them unless the launch splits the channels, launch_pre)
template <int PT, int WN>
__device__ __forceinline__ void layer3_max(const SAArgs &a, const float *A, int lda, int c2p, int row0, int b, int nc_lo = 0, int nc_hi = 1 << 30) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % WN, wp = wave / WN;
    const int KG = c2p / 16, NC = gp_round16(a.c3) / 16;
    if (nc_hi > NC) nc_hi = NC;
    const int pc0 = wp * PT;
    const int G = a.groupall ? PT : (a.ns >= 16 ? a.ns / 16 : 1);  // p-chunks per centre (nsample = 8: two centres per p-chunk)
    const bool half = !a.groupall && a.ns == 8;
    float *outb = a.out + (size_t)b * a.np * a.cout_total + a.cout_off;
    for (int ncb = nc_lo + wn; ncb < nc_hi; ncb += WN * 4) {
        int nc[4], nv = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            nc[i] = ncb + i * WN;
            nv += nc[i] < nc_hi;
        }
        f32x4 acc[4][PT];
        mfma_tile_n<PT>(nv, A, lda, pc0, a.w3, KG, NC, nc, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= nv) break;
            const int ch = nc[i] * 16 + 4 * (lane >> 4);
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.b3 + ch);
            f32x4 m = {0.f, 0.f, 0.f, 0.f};  // ReLU output is >= 0
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                f32x4 v = acc[i][p] + bv;
                m.x = fmaxf(m.x, v.x);
                m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z);
                m.w = fmaxf(m.w, v.w);
                if (half) {
                    // nsample = 8 (ClsMSG_CFG_Dense level 2, ClsMSG_CFG_Lighter level 3; pointnet2.py:47-78): rows 0-7 and 8-15 of the
                    // p-chunk are two neighbourhoods - pooled over the eight lanes of each half row, lanes 0 and 8 store
                    m.x = row8_max(m.x);
                    m.y = row8_max(m.y);
                    m.z = row8_max(m.z);
                    m.w = row8_max(m.w);
                    if ((lane & 7) == 0 && ch < a.c3) {
                        const int centre = (row0 + (pc0 + p) * 16) / 8 + ((lane >> 3) & 1);
                        if (centre < a.np) *reinterpret_cast<f32x4 *>(outb + (size_t)centre * a.cout_total + ch) = m;
                    }
                    m = f32x4{0.f, 0.f, 0.f, 0.f};
                } else if ((p + 1) % G == 0) {
                    m.x = row16_max(m.x);
                    m.y = row16_max(m.y);
                    m.z = row16_max(m.z);
                    m.w = row16_max(m.w);
                    if ((lane & 15) == 0 && ch < a.c3) {
                        if (a.groupall) {
                            // post-ReLU values are >= 0, so their bit patterns order like SIGNED integers; a signed max also
                            // keeps a stray -0.0f (0x80000000 = INT_MIN) below the zero-initialised buffer
                            int *o = reinterpret_cast<int *>(outb + ch);
                            atomicMax(o + 0, __float_as_int(m.x));
                            atomicMax(o + 1, __float_as_int(m.y));
                            atomicMax(o + 2, __float_as_int(m.z));
                            atomicMax(o + 3, __float_as_int(m.w));
                        } else {
                            const int centre = (row0 + (pc0 + p + 1 - G) * 16) / a.ns;
                            if (centre < a.np) *reinterpret_cast<f32x4 *>(outb + (size_t)centre * a.cout_total + ch) = m;
                        }
                    }
                    m = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
}

template <int P>
__global__ __launch_bounds__(256) void sa_mlp_kernel(SAArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, row0 = blockIdx.x * P;
    const int K0 = a.cin + 3, K0p = gp_round16(K0);
    const int c1p = gp_round16(a.c1), c2p = gp_round16(a.c2);
    const int lda = (K0p > c2p ? K0p : c2p) + GP_LD_PAD, ldb = c1p + GP_LD_PAD;
    float *A = lds, *Bf = lds + P * lda;
    const int nrows = a.groupall ? a.n : a.np * a.ns;

    const float *xyz = a.xyz + (size_t)b * a.n * 3;
    const float *fin = a.feats_in ? a.feats_in + (size_t)b * a.n * a.cin : nullptr;
    // ---- gather: xyz part (+ zero pad) : one thread per row
    for (int r = tid; r < P; r += 256) {
        int g = row0 + r;
        if (g >= nrows) g = nrows - 1;
        int j;
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (a.groupall) {
            j = g;
        } else {
            const int c = g / a.ns;
            j = a.idx[((size_t)b * a.np) * a.ns + g];
            const float *cp = a.new_xyz + ((size_t)b * a.np + c) * 3;
            cx = cp[0], cy = cp[1], cz = cp[2];
        }
        float *row = A + r * lda + a.cin;
        row[0] = xyz[j * 3 + 0] - cx;  // grouped_xyz -= new_xyz (pointnet2_utils.py:253)
        row[1] = xyz[j * 3 + 1] - cy;
        row[2] = xyz[j * 3 + 2] - cz;
        for (int k = K0; k < K0p; ++k) A[r * lda + k] = 0.f;
    }
    // ---- gather: feature part, float4 granules, consecutive threads -> consecutive granules of a row
    if (fin) {
        const int q4 = a.cin >> 2;  // cin is a multiple of 4 (96 / 256 / 512)
        for (int e = tid; e < P * q4; e += 256) {
            const int r = e / q4, q = e - r * q4;
            int g = row0 + r;
            if (g >= nrows) g = nrows - 1;
            const int j = a.groupall ? g : a.idx[((size_t)b * a.np) * a.ns + g];
            const f32x4 v = *reinterpret_cast<const f32x4 *>(fin + (size_t)j * a.cin + 4 * q);
            *reinterpret_cast<f32x4 *>(A + r * lda + 4 * q) = v;
        }
    }
    __syncthreads();
    dense_to_lds<P, true>(A, lda, a.w1, a.b1, K0, a.c1, Bf, ldb);
    __syncthreads();
    dense_to_lds<P, true>(Bf, ldb, a.w2, a.b2, a.c1, a.c2, A, lda);
    __syncthreads();
    // channel split (gridDim.z workgroups per row tile, launch<P> on small GroupAll batches): every workgroup has computed layers 1-2 for
    // its rows and takes its share of the last layer's output chunks - same MFMA order per output as the unsplit launch, bit for bit
    const int NC3 = gp_round16(a.c3) / 16, per = (NC3 + (int)gridDim.z - 1) / (int)gridDim.z;
    const int nc_lo = (int)blockIdx.z * per, nc_hi = nc_lo + per < NC3 ? nc_lo + per : NC3;
    if (nc_lo >= nc_hi) return;
    // a wave must own whole neighbourhoods for the in-register max: >= ns points per wave (GroupAll: any split, atomics combine)
    const int wn = pick_wn(nc_hi - nc_lo, P, a.groupall ? 16 : a.ns);
    if constexpr (P >= 64) {
        if (wn == 1) return layer3_max<P / 64, 1>(a, A, lda, c2p, row0, b, nc_lo, nc_hi);
    }
    if constexpr (P >= 32) {
        if (wn == 2) return layer3_max<P / 32, 2>(a, A, lda, c2p, row0, b, nc_lo, nc_hi);
    }
    layer3_max<P / 16, 4>(a, A, lda, c2p, row0, b, nc_lo, nc_hi);
}

// ------------------------------------------------------------------------------------------------------------------
// Hoisted first layer (exact algebra):  W1.[feat_j ; xyz_j - c] = W1f.feat_j + W1x.(xyz_j - c)
// The feature half depends on the SOURCE point only, so it is computed once per point (n rows) by point_linear_kernel
// instead of once per (centre, sample) pair (np*ns = 8..16 n rows); the xyz half is three FMAs per channel, applied while
// the neighbourhood is gathered.  Saves ~27 % of the encoder's MFMA work, halves the gather bytes and drops the K0-wide
// LDS buffer.  (The reference materialises the grouped tensor and convolves all of it, pointnet2_utils.py:246-258.)

// Z[row, 0:N] = X[row, 0:K] . W^T   (no bias, no activation), rows = B*n points, 64 rows per workgroup
__global__ __launch_bounds__(256) void point_linear_kernel(int M, int K, int N, const float *__restrict__ X, const float *__restrict__ Wp,
                                                           float *__restrict__ Z) {
    constexpr int P = 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * P, Kp = gp_round16(K), ld = Kp + GP_LD_PAD;
    const int q4 = K >> 2;
    for (int e = tid; e < P * q4; e += 256) {
        const int r = e / q4, q = e - r * q4;
        int g = row0 + r;
        if (g >= M) g = M - 1;
        *reinterpret_cast<f32x4 *>(lds + r * ld + 4 * q) = *reinterpret_cast<const f32x4 *>(X + (size_t)g * K + 4 * q);
    }
    if (Kp > K)
        for (int e = tid; e < P * (Kp - K); e += 256) lds[(e / (Kp - K)) * ld + K + e % (Kp - K)] = 0.f;
    __syncthreads();
    // every wave takes all 64 rows and every fourth channel chunk: a weight fragment feeds four MFMAs (with 16 rows per wave and
    // all chunks it fed one, and every wave streamed the whole weight set: the weight-stream-bound regime of DESIGN.md §4.2)
    const int KG = Kp / 16, NC = gp_round16(N) / 16;
    for (int ncb = wave; ncb < NC; ncb += 16) {
        int nc[4], nv = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            nc[i] = ncb + 4 * i;
            nv += nc[i] < NC;
        }
        f32x4 acc[4][4];
        mfma_tile_n<4>(nv, lds, ld, 0, Wp, KG, NC, nc, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= nv) break;
            const int ch = nc[i] * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = row0 + p * 16 + (lane & 15);
                if (row < M && ch < N) *reinterpret_cast<f32x4 *>(Z + (size_t)row * N + ch) = acc[i][p];
            }
        }
    }
}


// The same GEMM with the WEIGHTS STATIONARY IN REGISTERS, for the three shapes of the light encoder (rows x K -> N: B*512 x 96 -> 128,
// B*256 x 256 -> 256, B*128 x 512 -> 512).  The kernel above re-reads every weight fragment from the L2 for each 64-row tile and
// is latency-bound on that stream (0.34 / 0.51 / 0.66 of the MFMA peak on the three levels).  Here a wave keeps the A fragments of
// NCW output chunks x all KB k-blocks in registers for the whole launch (48 / 128 / 128 registers), the workgroup's four waves (and
// blockIdx.y) split the output channels, and persistent workgroups stream 16 * PT-row tiles of X through LDS: with PF the next tile
// is requested into registers before the current one is multiplied (the scheduling barrier keeps the requests there) and written to
// LDS behind it; WGS workgroups per CU fill each other's barrier bubbles.  The MFMA order per output (k-block major, jj inner) is
// the kernel above's: the results are bit-identical.
template <int KB, int NCW, int PT, bool PF, int WGS>
__global__ __launch_bounds__(256, WGS) void point_linear_ws_kernel(int M, int N, const float *__restrict__ X, const float *__restrict__ Wp,
                                                                 float *__restrict__ Z, int ntiles) {
    constexpr int K = 16 * KB, LD = K + GP_LD_PAD, R = 16 * PT;
    constexpr int PRE = R * (K / 4) / 256;  // f32x4 per thread and tile
    static_assert(R * (K / 4) % 256 == 0, "tile size");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NC = N / 16, c0 = (blockIdx.y * 4 + wave) * NCW;
    f32x4 wr[NCW][KB];
#pragma unroll
    for (int c = 0; c < NCW; ++c)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) wr[c][kb] = reinterpret_cast<const f32x4 *>(Wp)[((size_t)kb * NC + c0 + c) * 64 + lane];
    f32x4 pre[PF ? PRE : 1];
    auto request = [&](int t) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int e = tid + u * 256, r = e / (K / 4), q = e - r * (K / 4);
            int g = t * R + r;
            g = g < M ? g : M - 1;
            pre[u] = *reinterpret_cast<const f32x4 *>(X + (size_t)g * K + 4 * q);
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int e = tid + u * 256, r = e / (K / 4), q = e - r * (K / 4);
            *reinterpret_cast<f32x4 *>(lds + r * LD + 4 * q) = pre[u];
        }
    };
    int t = blockIdx.x;
    if constexpr (PF) {
        if (t < ntiles) request(t);
    }
    for (; t < ntiles; t += gridDim.x) {
        __syncthreads();  // every wave has read the previous tile
        if constexpr (PF) {
            deposit();
        } else {  // no register budget for a tile in flight: the co-resident workgroup covers this one's load
#pragma unroll
            for (int u = 0; u < PRE; ++u) {
                const int e = tid + u * 256, r = e / (K / 4), q = e - r * (K / 4);
                int g = t * R + r;
                g = g < M ? g : M - 1;
                *reinterpret_cast<f32x4 *>(lds + r * LD + 4 * q) = *reinterpret_cast<const f32x4 *>(X + (size_t)g * K + 4 * q);
            }
        }
        __syncthreads();
        if constexpr (PF) {
            if (t + (int)gridDim.x < ntiles) request(t + gridDim.x);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 acc[NCW][PT];
#pragma unroll
        for (int c = 0; c < NCW; ++c)
#pragma unroll
            for (int p = 0; p < PT; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *xb = lds + (lane & 15) * LD + 4 * (lane >> 4);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            f32x4 b[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) b[p] = *reinterpret_cast<const f32x4 *>(xb + p * 16 * LD + kb * 16);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int c = 0; c < NCW; ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[c][kb][jj], b[p][jj], acc[c][p], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < NCW; ++c) {
            const int ch = (c0 + c) * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int row = t * R + p * 16 + (lane & 15);
                if (row < M) *reinterpret_cast<f32x4 *>(Z + (size_t)row * N + ch) = acc[c][p];
            }
        }
    }
}

template <int KB, int NCW, int PT, bool PF, int WGS>
int launch_point_linear_ws(int rows, int n_out, const float *x, const float *wpack, float *z, hipStream_t st) {
    constexpr int K = 16 * KB, R = 16 * PT;
    constexpr size_t lds = (size_t)R * (K + GP_LD_PAD) * sizeof(float);
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(point_linear_ws_kernel<KB, NCW, PT, PF, WGS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return GP_ELAUNCH;
        done = true;
    }
    const int ntiles = (rows + R - 1) / R, nsplit = n_out / (64 * NCW);
    // persistent: WGS workgroups per CU over all channel splits, so that the splits of a tile run side by side and share its rows in the
    // L2 (a multiple of 8 in x keeps them on one XCD: the linear workgroup index is dealt round-robin over the eight)
    int gx = gp_num_cus() * WGS / nsplit;
    gx = ntiles < gx ? ntiles : gx;
    if (gx >= 8) gx &= ~7;
    hipLaunchKernelGGL((point_linear_ws_kernel<KB, NCW, PT, PF, WGS>), dim3(gx, nsplit), dim3(256), lds, st, rows, n_out, x, wpack, z, ntiles);
    return gp_launch_status();
}

struct SAPreArgs {
    int n, np, ns, c1, c2, c3, zstride, zoff;
    const float *xyz, *new_xyz, *z;  // z [b, n, zstride] or null (level 0: no input features)
    const int32_t *idx;
    const float *wxyz, *b1;          // wxyz [c1p][4] = (wx, wy, wz, 0) per channel; b1 [c1p]
    const float *w2, *b2, *w3, *b3;
    float *out;
    int cout_total, cout_off;
    int groupall;  // np = 1, the neighbourhood is every point of the cloud in order, no centring; tiles combine by atomic max
};

template <int P>
__global__ __launch_bounds__(256) void sa_pre_mlp_kernel(SAPreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, row0 = blockIdx.x * P;
    const int c1p = gp_round16(a.c1), c2p = gp_round16(a.c2);
    const int lda = c2p + GP_LD_PAD, ldb = c1p + GP_LD_PAD;
    float *A = lds, *Bf = lds + P * lda;
    float *dxyz = Bf + P * ldb;            // [P][4]
    int *src = reinterpret_cast<int *>(dxyz + P * 4);  // [P] source point of every row
    const int nrows = a.groupall ? a.n : a.np * a.ns;
    const float *xyz = a.xyz + (size_t)b * a.n * 3;
    for (int r = tid; r < P; r += 256) {
        int g = row0 + r;
        if (g >= nrows) g = nrows - 1;
        int j = g;
        float cx = 0.f, cy = 0.f, cz = 0.f;  // GroupAll keeps absolute coordinates (pointnet2_utils.py:281-289)
        if (!a.groupall) {
            const int c = g / a.ns;
            j = a.idx[((size_t)b * a.np) * a.ns + g];
            const float *cp = a.new_xyz + ((size_t)b * a.np + c) * 3;
            cx = cp[0], cy = cp[1], cz = cp[2];
        }
        dxyz[r * 4 + 0] = xyz[j * 3 + 0] - cx;  // grouped_xyz -= new_xyz (pointnet2_utils.py:253)
        dxyz[r * 4 + 1] = xyz[j * 3 + 1] - cy;
        dxyz[r * 4 + 2] = xyz[j * 3 + 2] - cz;
        dxyz[r * 4 + 3] = 0.f;
        src[r] = j;
    }
    __syncthreads();
    // ---- layer 1 while gathering: h1 = relu(Z[j] + wx*dx + wy*dy + wz*dz + b1)
    {
        const int q4 = c1p >> 2;
        const float *zb = a.z ? a.z + (size_t)b * a.n * a.zstride + a.zoff : nullptr;
        for (int e = tid; e < P * q4; e += 256) {
            const int r = e / q4, q = e - r * q4;
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dxyz + r * 4);
            f32x4 v = *reinterpret_cast<const f32x4 *>(a.b1 + 4 * q);
            if (zb && 4 * q < a.c1) v += *reinterpret_cast<const f32x4 *>(zb + (size_t)src[r] * a.zstride + 4 * q);
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(a.wxyz + (4 * q + 0) * 4);
            const f32x4 w1 = *reinterpret_cast<const f32x4 *>(a.wxyz + (4 * q + 1) * 4);
            const f32x4 w2 = *reinterpret_cast<const f32x4 *>(a.wxyz + (4 * q + 2) * 4);
            const f32x4 w3 = *reinterpret_cast<const f32x4 *>(a.wxyz + (4 * q + 3) * 4);
            v.x += w0.x * d.x + w0.y * d.y + w0.z * d.z;
            v.y += w1.x * d.x + w1.y * d.y + w1.z * d.z;
            v.z += w2.x * d.x + w2.y * d.y + w2.z * d.z;
            v.w += w3.x * d.x + w3.y * d.y + w3.z * d.z;
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<f32x4 *>(Bf + r * ldb + 4 * q) = v;
        }
    }
    __syncthreads();
    dense_to_lds<P, true>(Bf, ldb, a.w2, a.b2, a.c1, a.c2, A, lda);
    __syncthreads();
    SAArgs l3;  // layer 3 + max reuses the generic epilogue
    l3.n = a.n, l3.np = a.np, l3.ns = a.ns, l3.cin = 0, l3.c1 = a.c1, l3.c2 = a.c2, l3.c3 = a.c3;
    l3.xyz = nullptr, l3.feats_in = nullptr, l3.new_xyz = nullptr, l3.idx = nullptr;
    l3.w1 = l3.b1 = l3.w2 = l3.b2 = nullptr, l3.w3 = a.w3, l3.b3 = a.b3;
    l3.out = a.out, l3.cout_total = a.cout_total, l3.cout_off = a.cout_off, l3.groupall = a.groupall;
    // channel split (gridDim.z workgroups per row tile, launch_pre): every workgroup has computed layers 1-2 for its rows and takes its
    // share of the last layer's output chunks - same MFMA order per output, so the results are those of the unsplit launch, bit for bit
    const int NC3 = gp_round16(a.c3) / 16, per = (NC3 + (int)gridDim.z - 1) / (int)gridDim.z;
    const int nc_lo = (int)blockIdx.z * per, nc_hi = nc_lo + per < NC3 ? nc_lo + per : NC3;
    if (nc_lo >= nc_hi) return;
    const int wn = pick_wn(nc_hi - nc_lo, P, a.groupall ? 16 : a.ns);
    if constexpr (P >= 64) {
        if (wn == 1) return layer3_max<P / 64, 1>(l3, A, lda, c2p, row0, b, nc_lo, nc_hi);
    }
    if constexpr (P >= 32) {
        if (wn == 2) return layer3_max<P / 32, 2>(l3, A, lda, c2p, row0, b, nc_lo, nc_hi);
    }
    layer3_max<P / 16, 4>(l3, A, lda, c2p, row0, b, nc_lo, nc_hi);
}

template <int P>
int launch_pre(const SAPreArgs &a, int b, hipStream_t st) {
    const int c1p = gp_round16(a.c1), c2p = gp_round16(a.c2);
    const size_t lds = ((size_t)P * (c2p + c1p + 2 * GP_LD_PAD) + (size_t)P * 4 + P) * sizeof(float);
    if (lds > 160 * 1024) return GP_EINVAL;
    auto kern = sa_pre_mlp_kernel<P>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GP_ELAUNCH;
    }
    const int nrows = a.groupall ? a.n : a.np * a.ns;
    const int tiles = (nrows + P - 1) / P;
    // The GroupAll level of a SMALL batch (a tracking frame: 5 clouds = 20 tiles; one cloud: 4): every tile streams all of both weight
    // matrices (0.77 / 1.15 MB) through one CU while the others idle - 48 us per launch whatever the batch.  While the chip has CUs to spare,
    // 2 or 4 workgroups share a row tile: each recomputes layers 1-2 (a third of the work) and takes a half / quarter of layer 3's channels.
    int split = 1;
    if (a.groupall) {
        const int ncu = gp_num_cus();
        split = tiles * b * 4 <= ncu ? 4 : (tiles * b * 2 <= ncu ? 2 : 1);
    }
    hipLaunchKernelGGL(kern, dim3(tiles, b, split), dim3(256), lds, st, a);
    return gp_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Register-resident chain for the narrow first level (no input features; widths <= 64; ~1.6 M rows at B = 64).
// One WAVE owns one neighbourhood at a time and carries it through all three layers without LDS or barriers:
//   layer 1 is three FMAs per channel (the hoisted xyz half), computed directly in the MFMA B-operand layout;
//   the D fragment of v_mfma_f32_16x16x4_f32 (lane = point, 4 consecutive channels of chunk nc) IS the B fragment
//   the next layer needs for k-group nc, so activations never leave the registers;
//   all weights of the level (<= 14 KB) are preloaded as A fragments into registers once per wave.
// The tile-based kernel above spends its time on barriers and dependent global round trips here (measured 111 + 306 us
// for the two scales at B = 64 against ~10 + 45 us of MFMA work).
template <int C1, int C2, int C3, int NS>
__global__ __launch_bounds__(256, 3) void sa0_chain_kernel(SAPreArgs a, int ncentres_total) {
    constexpr int PT = NS / 16, Q1 = C1 / 16, Q2 = C2 / 16, Q3 = C3 / 16;
    const int lane = threadIdx.x & 63, pt = lane & 15, g = lane >> 4;
    const int wave_global = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = gridDim.x * 4;  // (scalar: the centre index and everything addressed through it stay on the scalar unit)
    // ---- per-lane constants: layer-1 rows for channels 16q + 4g + {0..3}; A fragments of layers 2 and 3; biases
    f32x4 w1[Q1][4], bb1[Q1];
#pragma unroll
    for (int q = 0; q < Q1; ++q) {
        bb1[q] = *reinterpret_cast<const f32x4 *>(a.b1 + 16 * q + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) w1[q][r] = *reinterpret_cast<const f32x4 *>(a.wxyz + (16 * q + 4 * g + r) * 4);
    }
    f32x4 w2[Q2][Q1], bb2[Q2], w3[Q3][Q2];
    float bb3[Q3];  // last layer runs transposed (lane = channel 16 n + pt): one bias per lane and chunk
#pragma unroll
    for (int n = 0; n < Q2; ++n) {
        bb2[n] = *reinterpret_cast<const f32x4 *>(a.b2 + 16 * n + 4 * g);
#pragma unroll
        for (int q = 0; q < Q1; ++q) w2[n][q] = reinterpret_cast<const f32x4 *>(a.w2)[((size_t)q * Q2 + n) * 64 + lane];
    }
#pragma unroll
    for (int n = 0; n < Q3; ++n) {
        bb3[n] = a.b3[16 * n + pt];
#pragma unroll
        for (int q = 0; q < Q2; ++q) w3[n][q] = reinterpret_cast<const f32x4 *>(a.w3)[((size_t)q * Q3 + n) * 64 + lane];
    }
    // ---- neighbourhood loop with a two-deep prefetch: idx of centre i+2 and xyz of centre i+1 are in flight while i runs
    auto centre_ok = [&](int c) { return c < ncentres_total; };
    auto load_idx = [&](int c, int (&j)[PT]) {
#pragma unroll
        for (int p = 0; p < PT; ++p) j[p] = centre_ok(c) ? a.idx[(size_t)c * NS + p * 16 + pt] : 0;
    };
    auto load_d = [&](int c, const int (&j)[PT], float (&d)[PT][3]) {
        const int cc = centre_ok(c) ? c : 0;
        const int bcl = cc / a.np;
        const float *xyz = a.xyz + (size_t)bcl * a.n * 3;
        const float *cp = a.new_xyz + (size_t)cc * 3;
        const float cx = cp[0], cy = cp[1], cz = cp[2];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            d[p][0] = xyz[j[p] * 3 + 0] - cx;  // grouped_xyz -= new_xyz (pointnet2_utils.py:253)
            d[p][1] = xyz[j[p] * 3 + 1] - cy;
            d[p][2] = xyz[j[p] * 3 + 2] - cz;
        }
    };
    int c = wave_global;
    int jn[PT], jnn[PT];
    float dcur[PT][3], dn[PT][3];
    load_idx(c, jn);
    load_d(c, jn, dcur);
    load_idx(c + nwaves, jn);
    for (; c < ncentres_total; c += nwaves) {
        load_d(c + nwaves, jn, dn);
        load_idx(c + 2 * nwaves, jnn);
        float res[Q3];  // running max over the neighbourhood's rows of the PRE-bias layer-3 output, per channel 16 n + pt
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const float dx = dcur[p][0], dy = dcur[p][1], dz = dcur[p][2];
            f32x4 h1[Q1], h2[Q2];
#pragma unroll
            for (int q = 0; q < Q1; ++q) {
                f32x4 v = bb1[q];
                v.x += w1[q][0].x * dx + w1[q][0].y * dy + w1[q][0].z * dz;
                v.y += w1[q][1].x * dx + w1[q][1].y * dy + w1[q][1].z * dz;
                v.z += w1[q][2].x * dx + w1[q][2].y * dy + w1[q][2].z * dz;
                v.w += w1[q][3].x * dx + w1[q][3].y * dy + w1[q][3].z * dz;
                h1[q] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
            }
            // every output chunk of a layer has its own accumulator and the chunks are interleaved k-step by k-step: consecutive
            // MFMAs never wait on each other's result (40-cycle dependent latency against a 32-cycle issue interval)
            {
                f32x4 acc[Q2];
#pragma unroll
                for (int n = 0; n < Q2; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < Q1; ++q)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int n = 0; n < Q2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[n][q][jj], h1[q][jj], acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < Q2; ++n) {
                    const f32x4 v = acc[n] + bb2[n];
                    h2[n] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                }
            }
            {
                // last layer TRANSPOSED (activations as the A operand): lane = channel, registers x lane groups = the 16 rows
                f32x4 acc[Q3];
#pragma unroll
                for (int n = 0; n < Q3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < Q2; ++q)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int n = 0; n < Q3; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(h2[q][jj], w3[n][q][jj], acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < Q3; ++n) {
                    const float m = points16_max_t(acc[n]);
                    res[n] = p == 0 ? m : fmaxf(res[n], m);
                }
            }
        }
        // max_i relu(x_i + b) = relu(max_i x_i + b): bias and ReLU once per channel, after the pooling (exact: rounding is monotone)
        float *o = a.out + (size_t)c * a.cout_total + a.cout_off;
#pragma unroll
        for (int n = 0; n < Q3; ++n)
            if (g == 0) o[16 * n + pt] = fmaxf(res[n] + bb3[n], 0.f);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            dcur[p][0] = dn[p][0], dcur[p][1] = dn[p][1], dcur[p][2] = dn[p][2];
            jn[p] = jnn[p];
        }
    }
}

// Same register-resident chain for a level WITH input features (hoisted first layer: h1 = relu(Z[j] + W1x.dxyz + b1)).
// The layer-2/3 weights (48-72 KB for the light config's second level) no longer fit the register file, so they are
// copied ONCE per workgroup into LDS in A-fragment order and every wave streams its fragments from there
// (ds_read_b128, 1 KB per wave-instruction, lane-linear = conflict free) while it walks its own neighbourhoods:
// no L2 weight traffic after the prologue, no LDS activation traffic, no barriers in the loop.
template <int C1, int C2, int C3, int NS>
__global__ __launch_bounds__(256, 3) void sa_chain_lds_kernel(SAPreArgs a, int ncentres_total) {
    constexpr int PT = NS / 16, Q1 = C1 / 16, Q2 = (C2 + 15) / 16, Q3 = C3 / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4 *w2l = reinterpret_cast<f32x4 *>(lds);           // [Q1][Q2][64]
    f32x4 *w3l = w2l + Q1 * Q2 * 64;                        // [Q2][Q3][64]
    f32x4 *w1l = w3l + Q2 * Q3 * 64;                        // [C1] rows (wx, wy, wz, b1)
    const int tid = threadIdx.x, lane = tid & 63, pt = lane & 15, g = lane >> 4;
    for (int e = tid; e < Q1 * Q2 * 64; e += 256) w2l[e] = reinterpret_cast<const f32x4 *>(a.w2)[e];
    for (int e = tid; e < Q2 * Q3 * 64; e += 256) w3l[e] = reinterpret_cast<const f32x4 *>(a.w3)[e];
    for (int e = tid; e < C1; e += 256) {
        f32x4 w = *reinterpret_cast<const f32x4 *>(a.wxyz + e * 4);
        w.w = a.b1[e];
        w1l[e] = w;
    }
    f32x4 bb2[Q2];
    float bb3[Q3];  // the last layer runs transposed (lane = channel 16 n + pt, see points16_max_t): one bias per lane and chunk
#pragma unroll
    for (int n = 0; n < Q2; ++n) bb2[n] = *reinterpret_cast<const f32x4 *>(a.b2 + 16 * n + 4 * g);
#pragma unroll
    for (int n = 0; n < Q3; ++n) bb3[n] = a.b3[16 * n + pt];
    __syncthreads();
    const int wave_global = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = gridDim.x * 4;  // (scalar, as in sa0_chain_kernel)
    // The wave walks 16-row chunks: iteration `it` is chunk p = it % PT of the wave's (it / PT)-th neighbourhood, so the
    // per-iteration register footprint is one chunk whatever the neighbourhood size (the running max lives in `res`).
    const int my_centres = wave_global < ncentres_total ? (ncentres_total - wave_global + nwaves - 1) / nwaves : 0;
    const int nits = my_centres * PT;
    auto chunk_row0 = [&](int it, int &c) {
        c = wave_global + (it / PT) * nwaves;
        return (size_t)c * NS + (size_t)(it % PT) * 16;
    };
    auto load_idx = [&](int it) {
        int c;
        const size_t r0 = chunk_row0(it, c);
        return it < nits ? a.idx[r0 + pt] : 0;
    };
    // operands of one chunk: xyz deltas and the gathered rows of Z (this lane's channels 16q + 4g + 0..3)
    auto load_ops = [&](int it, int j, float (&d)[3], f32x4 (&zz)[Q1]) {
        int c;
        chunk_row0(it, c);
        const int cc = it < nits ? c : 0;
        const int bcl = cc / a.np;
        const float *xyz = a.xyz + (size_t)bcl * a.n * 3;
        const float *zb = a.z + (size_t)bcl * a.n * a.zstride + a.zoff;
        const float *cp = a.new_xyz + (size_t)cc * 3;
        d[0] = xyz[j * 3 + 0] - cp[0];  // grouped_xyz -= new_xyz (pointnet2_utils.py:253)
        d[1] = xyz[j * 3 + 1] - cp[1];
        d[2] = xyz[j * 3 + 2] - cp[2];
#pragma unroll
        for (int q = 0; q < Q1; ++q) zz[q] = *reinterpret_cast<const f32x4 *>(zb + (size_t)j * a.zstride + 16 * q + 4 * g);
    };
    int jn, jnn;
    float dcur[3], dn[3];
    f32x4 zcur[Q1], zn[Q1];
    jn = load_idx(0);
    load_ops(0, jn, dcur, zcur);
    jn = load_idx(1);
    float res[Q3];  // running max over the neighbourhood's rows of the pre-bias layer-3 output, per channel 16 n + pt
    // The weight fragments are read from LDS as ONE software-pipelined stream per 16-row chunk: layer-2 groups (two output chunks of
    // one k-group: 8 MFMAs) in order (n0, q), then layer-3 groups (four output chunks of one k-group: 16 MFMAs); the fragments of
    // group g + 1 are requested before the MFMAs of group g, and the scheduling barriers keep them there (hipcc otherwise sinks every
    // ds_read next to its first use: an LDS round trip in front of every 8 MFMAs).  The first group of the NEXT chunk is requested
    // under the last group of this one (the weights in LDS never change).
    constexpr int H2 = (Q2 + 1) / 2, G2 = H2 * Q1, G3 = (Q3 / 4) * Q2;
    f32x4 w2b[2][2], w3b[2][4];
    auto ld2 = [&](int g, f32x4 (&w)[2], int lo) {
        const int n0 = 2 * (g / Q1), q = g % Q1;
#pragma unroll
        for (int u = 0; u < 2; ++u) w[u] = (n0 + u < Q2) ? w2l[(q * Q2 + n0 + u) * 64 + lo] : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto ld3 = [&](int g, f32x4 (&w)[4], int lo) {
        const int n0 = 4 * (g / Q2), q = g % Q2;
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = w3l[(q * Q3 + n0 + u) * 64 + lo];
    };
    ld2(0, w2b[0], lane);
#pragma unroll 1
    for (int it = 0; it < nits; ++it) {
        load_ops(it + 1, jn, dn, zn);
        jnn = load_idx(it + 2);
        // the weight fragments are loop-invariant LDS reads: launder the lane offset so hipcc keeps them as streamed
        // ds_read_b128 inside the loop instead of hoisting all (Q1*Q2 + Q2*Q3) fragments into registers (spills)
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int p = it % PT;
        {
            const float dx = dcur[0], dy = dcur[1], dz = dcur[2];
            f32x4 h1[Q1], h2[Q2];
#pragma unroll
            for (int q = 0; q < Q1; ++q) {
                const int g4 = (lo >> 4) * 4;
                const f32x4 r0 = w1l[16 * q + g4 + 0], r1 = w1l[16 * q + g4 + 1], r2 = w1l[16 * q + g4 + 2], r3 = w1l[16 * q + g4 + 3];
                f32x4 v = zcur[q];
                v.x += (r0.x * dx + r0.y * dy + r0.z * dz) + r0.w;
                v.y += (r1.x * dx + r1.y * dy + r1.z * dz) + r1.w;
                v.z += (r2.x * dx + r2.y * dy + r2.z * dz) + r2.w;
                v.w += (r3.x * dx + r3.y * dy + r3.z * dz) + r3.w;
                h1[q] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
            }
            // layer 2: two output chunks in flight (independent accumulators hide the 40-cycle dependent MFMA latency)
            f32x4 acc2[2];
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                const int n0 = 2 * (g / Q1), q = g % Q1;
                if (g + 1 < G2)
                    ld2(g + 1, w2b[(g + 1) & 1], lo);
                else
                    ld3(0, w3b[0], lo);
                __builtin_amdgcn_sched_barrier(0);
                if (q == 0) acc2[0] = acc2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (n0 + u < Q2) acc2[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2b[g & 1][u][jj], h1[q][jj], acc2[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q == Q1 - 1) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (n0 + u < Q2) {
                            f32x4 v = acc2[u] + bb2[n0 + u];
                            h2[n0 + u] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                        }
                }
            }
            // layer 3 + running max over the neighbourhood's chunks: four output chunks in flight
            f32x4 acc3[4];
#pragma unroll
            for (int g = 0; g < G3; ++g) {
                const int n0 = 4 * (g / Q2), q = g % Q2;
                if (g + 1 < G3)
                    ld3(g + 1, w3b[(g + 1) & 1], lo);
                else
                    ld2(0, w2b[0], lo);  // first group of the next chunk
                __builtin_amdgcn_sched_barrier(0);
                if (q == 0) acc3[0] = acc3[1] = acc3[2] = acc3[3] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc3[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(h2[q][jj], w3b[g & 1][u][jj], acc3[u], 0, 0, 0);  // transposed
                __builtin_amdgcn_sched_barrier(0);
                if (q == Q2 - 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float m = points16_max_t(acc3[u]);
                        res[n0 + u] = p == 0 ? m : fmaxf(res[n0 + u], m);
                    }
                }
            }
        }
        if (p == PT - 1) {
            // max_i relu(x_i + b) = relu(max_i x_i + b): bias and ReLU once per channel, after the pooling (exact: rounding is monotone)
            const int c = wave_global + (it / PT) * nwaves;
            float *o = a.out + (size_t)c * a.cout_total + a.cout_off;
#pragma unroll
            for (int n = 0; n < Q3; ++n)
                if (g == 0) o[16 * n + pt] = fmaxf(res[n] + bb3[n], 0.f);
        }
        dcur[0] = dn[0], dcur[1] = dn[1], dcur[2] = dn[2];
        jn = jnn;
#pragma unroll
        for (int q = 0; q < Q1; ++q) zcur[q] = zn[q];
    }
}

template <int C1, int C2, int C3, int NS>
int launch_chain_lds(const SAPreArgs &a, int b, hipStream_t st) {
    constexpr int Q1 = C1 / 16, Q2 = (C2 + 15) / 16, Q3 = C3 / 16;
    static_assert(Q3 % 4 == 0, "layer-3 width must be a multiple of 64");
    const size_t lds = ((size_t)(Q1 * Q2 + Q2 * Q3) * 64 + C1) * sizeof(f32x4);
    auto kern = sa_chain_lds_kernel<C1, C2, C3, NS>;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GP_ELAUNCH;
        done = true;
    }
    const int ncentres = b * a.np;
    // persistent workgroups per CU: as many as the LDS holds, at most three (132-152 registers: three waves per SIMD; a third wave's VALU
    // work - layer 1, bias + ReLU, pooling - runs under the other two's MFMAs)
    const int fit = (int)((160 * 1024) / lds), per_cu = fit < 1 ? 1 : (fit > 3 ? 3 : fit);
    int blocks = (ncentres + 3) / 4;
    if (blocks > gp_num_cus() * per_cu) blocks = gp_num_cus() * per_cu;  // persistent: weights are staged once per workgroup
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, a, ncentres);
    return gp_launch_status();
}

// Chain kernel for a level whose layer-3 weights do not fit LDS (light config level 2: 128 -> 196 -> 256):
//   * 8 waves per workgroup, each wave walks its own neighbourhoods 16 rows at a time, activations in registers;
//   * layer-2 weights (Q1*Q2 KB) are LDS-resident; layer-3 weights stream through a 3-slot LDS ring, one k-group slice
//     (Q3 KB) per step, shared by all 8 waves: slice g+2 is written while slice g is multiplied, ONE barrier per step;
//   * every wave runs the same number of iterations (idle ones compute on clamped rows and store nothing) so the
//     barriers line up.
// L2 traffic per 128 rows: one sweep of the layer-3 weights (208 KB) instead of one sweep of both layers per 32 rows.
// SPREAD (hidden-layer layout GP_SA_TAIL_SPREAD, genpose_hip.h): the r = C2 % 16 channels of the last, partly filled 16-channel block
// sit at positions 4 (c % 4) + c / 4, i.e. in k-steps jj < ceil(r / 4) of all four lane groups, so layer 3 skips the k-steps of
// that block that only multiply padding (196 channels: one MFMA instead of four, -5.8 % of layer 3).
// out[r][0 .. 4 w4) = 0 for the rows of a [rows][ld] tensor: the zeroed output the SPLITP form below combines into
__global__ __launch_bounds__(256) void zero_columns_kernel(float *out, int ld, int w4, int total) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < total) *reinterpret_cast<f32x4 *>(out + (size_t)(e / w4) * ld + 4 * (e % w4)) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// SPLITP (small batches: fewer neighbourhoods than the chip has wave slots for): the unit of work is one 16-row CHUNK of a neighbourhood
// instead of the neighbourhood - twice the waves, half the iterations; a unit applies bias and ReLU to its own maximum and combines with
// the neighbourhood's other chunk through an integer atomic max (values >= 0: the order of the bit patterns) into the ZEROED output.
template <int C1, int C2, int C3, int NS, bool SPREAD, bool SPLITP = false>
__global__ __launch_bounds__(512) void sa_chain_ring_kernel(SAPreArgs a, int ncentres_total) {
    constexpr int PT = NS / 16, Q1 = C1 / 16, Q2 = (C2 + 15) / 16, Q3 = C3 / 16, NWV = 8, NTH = 512;
    constexpr int TAIL_JJ = (SPREAD && C2 % 16) ? (C2 % 16 + 3) / 4 : 4;  // k-steps of the last k-block that carry channels
    static_assert(Q3 % 4 == 0 && (Q3 * 64) % NTH == 0, "layer-3 slice must split evenly over the workgroup");
    constexpr int SLICE = Q3 * 64;          // f32x4 per k-group slice of layer 3
    constexpr int PER_T = SLICE / NTH;      // f32x4 each thread moves per slice
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4 *w2l = reinterpret_cast<f32x4 *>(lds);   // [Q1][Q2][64] resident
    f32x4 *ring = w2l + Q1 * Q2 * 64;               // [3][Q3][64]
    f32x4 *w1l = ring + 3 * SLICE;                  // [C1] rows (wx, wy, wz, b1)
    float *b2l = reinterpret_cast<float *>(w1l + C1);  // [16 Q2] layer-2 bias, [C3] layer-3 bias: read where they are used, from LDS -
    float *b3l = b2l + 16 * Q2;                         // a global read there waits (in-order vmcnt) for every operand request in flight
    const int tid = threadIdx.x, lane = tid & 63, pt = lane & 15, g = lane >> 4;
    for (int e = tid; e < Q1 * Q2 * 64; e += NTH) w2l[e] = reinterpret_cast<const f32x4 *>(a.w2)[e];
    for (int e = tid; e < C1; e += NTH) {
        f32x4 w = *reinterpret_cast<const f32x4 *>(a.wxyz + e * 4);
        w.w = a.b1[e];
        w1l[e] = w;
    }
    for (int e = tid; e < 16 * Q2; e += NTH) b2l[e] = a.b2[e];
    for (int e = tid; e < C3; e += NTH) b3l[e] = a.b3[e];
    const f32x4 *w3g = reinterpret_cast<const f32x4 *>(a.w3);  // [Q2][Q3][64]: slice q = w3g + q*SLICE
    // ring prologue: slices 0 and 1 into slots 0 and 1; slice 2 held in registers
    f32x4 hold[PER_T];
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
        ring[0 * SLICE + tid + u * NTH] = w3g[0 * SLICE + tid + u * NTH];
        ring[1 * SLICE + tid + u * NTH] = w3g[(1 % Q2) * SLICE + tid + u * NTH];
        hold[u] = w3g[(2 % Q2) * SLICE + tid + u * NTH];
    }
    __syncthreads();
    // the wave's index as a SCALAR (the centre index and everything addressed through it stay on the scalar unit: -7 % VALU instructions,
    // -2.8 % time at NS = 32); at NS = 16 the same change measured +1.3 % (another schedule of the same loop), so that form keeps the vector index
    const int wave_in_wg = NS == 16 ? (tid >> 6) : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_global = blockIdx.x * NWV + wave_in_wg, nwaves = gridDim.x * NWV;
    const int nunits = SPLITP ? ncentres_total * PT : ncentres_total;   // what the waves share out: chunks or whole neighbourhoods
    const int my_units = wave_global < nunits ? (nunits - wave_global + nwaves - 1) / nwaves : 0;
    const int nits = SPLITP ? my_units : my_units * PT;
    const int nits_wg = ((nunits + nwaves - 1) / nwaves) * (SPLITP ? 1 : PT);  // uniform over the grid: barrier counts match
    auto chunk_row0 = [&](int it, int &c) {
        if constexpr (SPLITP) {
            const int u = wave_global + it * nwaves;
            c = u / PT;
            return (size_t)c * NS + (size_t)(u % PT) * 16;
        }
        c = wave_global + (it / PT) * nwaves;
        return (size_t)c * NS + (size_t)(it % PT) * 16;
    };
    auto load_idx = [&](int it) {
        int c;
        const size_t r0 = chunk_row0(it, c);
        return it < nits ? a.idx[r0 + pt] : 0;
    };
    auto load_ops = [&](int it, int j, float (&d)[3], f32x4 (&zz)[Q1]) {
        int c;
        chunk_row0(it, c);
        const int cc = it < nits ? c : 0;
        const int bcl = cc / a.np;
        const float *xyz = a.xyz + (size_t)bcl * a.n * 3;
        const float *zb = a.z + (size_t)bcl * a.n * a.zstride + a.zoff;
        const float *cp = a.new_xyz + (size_t)cc * 3;
        d[0] = xyz[j * 3 + 0] - cp[0];  // grouped_xyz -= new_xyz (pointnet2_utils.py:253)
        d[1] = xyz[j * 3 + 1] - cp[1];
        d[2] = xyz[j * 3 + 2] - cp[2];
#pragma unroll
        for (int q = 0; q < Q1; ++q) zz[q] = *reinterpret_cast<const f32x4 *>(zb + (size_t)j * a.zstride + 16 * q + 4 * g);
    };
    // register budget (256 at two waves per SIMD): the operands of the NEXT chunk are requested into the same registers
    // right after layer 1 has consumed the current ones (the ~40 k cycles of layers 2-3 cover the latency); biases are
    // re-read from LDS where they are used; the running max of a two-chunk neighbourhood is one register per output chunk (`res`).
    int jn;
    float dcur[3];
    f32x4 zcur[Q1];
    jn = load_idx(0);
    load_ops(0, jn, dcur, zcur);
    jn = load_idx(1);
    int gstep = 0;  // global ring step: slice gstep % Q2 sits in slot gstep % 3
    f32x4 wpre[4];  // first fragment group of the upcoming ring step
    float res[Q3];  // running max of the pre-bias layer-3 output over the chunks of a neighbourhood, per channel 16 n + pt
#pragma unroll
    for (int u = 0; u < 4; ++u) wpre[u] = ring[u * 64 + lane];
#pragma unroll 1
    for (int it = 0; it < nits_wg; ++it) {
        int lo = lane;
        asm volatile("" : "+v"(lo));  // keep the LDS weight reads inside the loop (see sa_chain_lds_kernel)
        const int p = it % PT;
        const float dx = dcur[0], dy = dcur[1], dz = dcur[2];
        f32x4 h1[Q1], h2[Q2];
#pragma unroll
        for (int q = 0; q < Q1; ++q) {
            const int g4 = (lo >> 4) * 4;
            const f32x4 r0 = w1l[16 * q + g4 + 0], r1 = w1l[16 * q + g4 + 1], r2 = w1l[16 * q + g4 + 2], r3 = w1l[16 * q + g4 + 3];
            f32x4 v = zcur[q];
            v.x += (r0.x * dx + r0.y * dy + r0.z * dz) + r0.w;
            v.y += (r1.x * dx + r1.y * dy + r1.z * dz) + r1.w;
            v.z += (r2.x * dx + r2.y * dy + r2.z * dz) + r2.w;
            v.w += (r3.x * dx + r3.y * dy + r3.z * dz) + r3.w;
            h1[q] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
        }
        load_ops(it + 1, jn, dcur, zcur);
        jn = load_idx(it + 2);
        // ---- layer 2 from the resident weights, two output chunks in flight
#pragma unroll
        for (int n0 = 0; n0 < Q2; n0 += 2) {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int q = 0; q < Q1; ++q) {
                f32x4 wf[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) wf[u] = (n0 + u < Q2) ? w2l[(q * Q2 + n0 + u) * 64 + lo] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (n0 + u < Q2) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][jj], h1[q][jj], acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (n0 + u < Q2) {
                    f32x4 v = acc[u] + *reinterpret_cast<const f32x4 *>(b2l + 16 * (n0 + u) + 4 * (lo >> 4));
                    h2[n0 + u] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                }
        }
        // ---- layer 3 through the ring: Q2 steps, all Q3 output chunks accumulate across the steps.
        // Fragment groups are requested one group ahead; the first group of the NEXT step is requested before this step's barrier
        // (its slice was written two steps ago and published by the previous barrier), so no step starts on an LDS round trip.
        f32x4 acc3[Q3];
#pragma unroll
        for (int n = 0; n < Q3; ++n) acc3[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < Q2; ++q) {
            // slice gstep+2 (held in registers since the previous step) -> its slot, which was last read in step gstep-1
            {
                f32x4 *dst = ring + ((gstep + 2) % 3) * SLICE;
#pragma unroll
                for (int u = 0; u < PER_T; ++u) dst[tid + u * NTH] = hold[u];
                const f32x4 *src = w3g + ((gstep + 3) % Q2) * SLICE;
#pragma unroll
                for (int u = 0; u < PER_T; ++u) hold[u] = src[tid + u * NTH];
            }
            const f32x4 *slot = ring + (gstep % 3) * SLICE, *nslot = ring + ((gstep + 1) % 3) * SLICE;
#pragma unroll
            for (int n0 = 0; n0 < Q3; n0 += 4) {
                f32x4 wf[4], wn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) wf[u] = wpre[u];
#pragma unroll
                for (int u = 0; u < 4; ++u) wn[u] = (n0 + 4 < Q3) ? slot[(n0 + 4 + u) * 64 + lo] : nslot[u * 64 + lo];
#pragma unroll
                for (int jj = 0; jj < (q == Q2 - 1 ? TAIL_JJ : 4); ++jj)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc3[n0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(h2[q][jj], wf[u][jj], acc3[n0 + u], 0, 0, 0);  // transposed
#pragma unroll
                for (int u = 0; u < 4; ++u) wpre[u] = wn[u];
            }
            ++gstep;
            // (the fence-free s_barrier of trunk_chain.h measured equal here: two waves per SIMD cover the fence.  What the 13 barriers of an
            // iteration cost in all: without them - wrong results, timing only - the kernel runs 5.4 % (NS = 32) / 3.7 % (NS = 16) faster, round 5)
            __syncthreads();
        }
        {
            int c;
            chunk_row0(it, c);
            float *o = a.out + (size_t)(it < nits ? c : 0) * a.cout_total + a.cout_off;
            // layer 3 ran transposed: lane = channel 16 n + pt, the 16 rows are the registers x lane groups (points16_max_t);
            // max_i relu(x_i + b) = relu(max_i x_i + b), so bias and ReLU come once per channel after the pooling
#pragma unroll
            for (int n = 0; n < Q3; ++n) {
                const float m = points16_max_t(acc3[n]);
                if constexpr (SPLITP) {
                    // the chunk's own relu(max + b) (>= 0: its bit pattern orders like a signed integer), combined with the neighbourhood's
                    // other chunk in the zeroed output: max_chunks relu(max_chunk + b) = relu(max_all + b)
                    const float v = fmaxf(m + b3l[16 * n + (lo & 15)], 0.f);
                    if (g == 0 && it < nits) atomicMax(reinterpret_cast<int *>(o + 16 * n + pt), __float_as_int(v));
                } else {
                    res[n] = p == 0 ? m : fmaxf(res[n], m);  // running max over the neighbourhood's chunks: one register per chunk
                    if (p == PT - 1 && g == 0 && it < nits) o[16 * n + pt] = fmaxf(res[n] + b3l[16 * n + (lo & 15)], 0.f);
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// GroupAll level (128 points x [256 hoisted + xyz] -> C2 -> 512 per cloud and scale) as a register chain with BOTH weight matrices through
// an LDS ring: one 8-wave workgroup owns one CLOUD at a time (8 waves x 16 rows = its 128 points), so that
//   * a pass over the weights (0.77 / 1.15 MB) serves 128 rows instead of the 32 of the tile kernel, which is bound by that stream;
//   * the pooling over the cloud needs no atomics: a wave pools its 16 rows in registers (last layer transposed, lane = channel), the
//     eight waves meet in LDS once per cloud.
// Ring: three slots, one 16-wide k-block of a layer per step (layer 2: Q2 fragments; layer 3: 16 of its 32 output chunks, two halves
// of Q2 steps each, which keeps the accumulators at 64 registers), one barrier per step, the slice two steps ahead held in registers -
// the scheme of sa_chain_ring_kernel; the stream runs on from cloud to cloud.  Two waves per SIMD, compiler-scheduled: one wave's
// VALU work (layer 1, bias + ReLU, pooling) runs under the other's MFMAs (profiles/r3_sa_ring32_experiment.txt).
template <int C1, int C2, int C3>
__global__ __launch_bounds__(512) void sa_groupall_ring_kernel(SAPreArgs a, int nclouds) {
    constexpr int Q1 = C1 / 16, Q2 = C2 / 16, Q3 = C3 / 16, NWV = 8, NTH = 512;
    static_assert(C1 % 16 == 0 && C2 % 16 == 0 && Q3 == 32 && Q2 % 4 == 0, "light encoder's GroupAll shapes");
    constexpr int SLOTF = Q2 > 16 ? Q2 : 16;     // fragments per ring slot
    constexpr int SLOT = SLOTF * 64;             // f32x4 per slot
    constexpr int PER_T = (SLOT + NTH - 1) / NTH;
    constexpr int NSL = Q1 + 2 * Q2;             // slices per cloud: layer 2 k-blocks, then layer 3 (half 0, half 1) k-blocks
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NRS = 4;                                 // ring slots
    f32x4 *ring = reinterpret_cast<f32x4 *>(lds);          // [NRS][SLOT]
    f32x4 *w1l = ring + NRS * SLOT;                        // [C1] rows (wx, wy, wz, b1)
    float *pool = reinterpret_cast<float *>(w1l + C1);     // [NWV][C3] per-wave maxima of the pre-bias layer-3 output
    float *b2l = pool + NWV * C3;                          // [C2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), pt = lane & 15;
    const f32x4 *w2g = reinterpret_cast<const f32x4 *>(a.w2), *w3g = reinterpret_cast<const f32x4 *>(a.w3);
    // slice si of the per-cloud stream: base pointer and fragment count
    auto slice = [&](int si, int &nfrag) -> const f32x4 * {
        si %= NSL;
        if (si < Q1) {
            nfrag = Q2;
            return w2g + (size_t)si * Q2 * 64;
        }
        const int t = si - Q1, half = t / Q2, kb = t - half * Q2;
        nfrag = 16;
        return w3g + ((size_t)kb * Q3 + half * 16) * 64;
    };
    // a slice is requested into registers in step g and written to its slot in step g + 2 (two register sets): its L2 latency has two
    // steps (~2 us) to pass before anything waits for it
    f32x4 hold[2][PER_T];
    auto request = [&](int set, int si) {
        int nf;
        const f32x4 *src = slice(si, nf);
#pragma unroll
        for (int u = 0; u < PER_T; ++u) {
            const int e = tid + u * NTH;
            hold[set][u] = src[e < nf * 64 ? e : nf * 64 - 1];
        }
    };
    auto deposit = [&](int set, int slot) {
#pragma unroll
        for (int u = 0; u < PER_T; ++u) {
            const int e = tid + u * NTH;
            if (e < SLOT) ring[slot * SLOT + e] = hold[set][u];
        }
    };
    for (int e = tid; e < C1; e += NTH) {
        f32x4 w = *reinterpret_cast<const f32x4 *>(a.wxyz + e * 4);
        w.w = a.b1[e];
        w1l[e] = w;
    }
    for (int e = tid; e < C2; e += NTH) b2l[e] = a.b2[e];
    const float bias3 = a.b3[tid];
    request(0, 0);
    deposit(0, 0);
    request(0, 1);
    deposit(0, 1);
    request(0, 2);  // slices 2 and 3 travel in the two register sets: deposited in steps 0 and 1
    request(1, 3);
    __syncthreads();
    int gstep = 0;  // global ring step: slice gstep % NSL sits in slot gstep % NRS; slices gstep + 2, gstep + 3 are in `hold`
    f32x4 wpre[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wpre[u] = ring[u * 64 + lane];
    // one ring step over `NF` fragments (chunks n = 0 .. NF-1 of this slice), the fragments requested one group of four ahead and the first
    // group of the next step before the barrier
    // (the register set must be a compile-time index: every step is written for an even and an odd gstep)
    auto step_begin = [&](auto par) {
        constexpr int set = decltype(par)::value;
        deposit(set, (gstep + 2) % NRS);  // slice gstep + 2 (requested two steps ago) -> its slot, last read in step gstep - 2
        request(set, gstep + 4);
    };
#pragma unroll 1
    for (int cloud = blockIdx.x; cloud < nclouds; cloud += gridDim.x) {
        // the per-channel operands in LDS do not change from cloud to cloud: an opaque lane group keeps their reads INSIDE the loop
        // (hoisted out of it they are 90 registers that spill)
        int g = lane >> 4;
        asm volatile("" : "+v"(g));
        // ---- layer 1 (hoisted) is computed k-block by k-block inside layer 2's steps (sixteen fragments held at once, beside the
        // accumulators, spill): row = point 16 wave + pt of the cloud, absolute coordinates (GroupAll: no centring); its z values are
        // requested two steps ahead
        const int row = wave * 16 + pt;
        const float *xyzp = a.xyz + ((size_t)cloud * a.n + row) * 3;
        const float dx = xyzp[0], dy = xyzp[1], dz = xyzp[2];
        const float *zb = a.z + ((size_t)cloud * a.n + row) * a.zstride + a.zoff + 4 * g;
        f32x4 zq[4];
        zq[0] = *reinterpret_cast<const f32x4 *>(zb);
        zq[1] = *reinterpret_cast<const f32x4 *>(zb + 16);
        zq[2] = *reinterpret_cast<const f32x4 *>(zb + 32);
        auto layer1 = [&](int q) {
            const f32x4 r0 = w1l[16 * q + 4 * g + 0], r1 = w1l[16 * q + 4 * g + 1], r2 = w1l[16 * q + 4 * g + 2], r3 = w1l[16 * q + 4 * g + 3];
            f32x4 v = zq[q % 4];
            v.x += (r0.x * dx + r0.y * dy + r0.z * dz) + r0.w;
            v.y += (r1.x * dx + r1.y * dy + r1.z * dz) + r1.w;
            v.z += (r2.x * dx + r2.y * dy + r2.z * dz) + r2.w;
            v.w += (r3.x * dx + r3.y * dy + r3.z * dz) + r3.w;
            return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
        };
        f32x4 h1q = layer1(0);  // k-block q + 1 is prepared while k-block q is multiplied
        // ---- layer 2: Q1 ring steps, all Q2 output chunks accumulate across them
        f32x4 h2[Q2];
#pragma unroll
        for (int n = 0; n < Q2; ++n) h2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < Q1; ++q) {
            if (q & 1) step_begin(std::integral_constant<int, 1>{}); else step_begin(std::integral_constant<int, 0>{});
            if (q + 3 < Q1) zq[(q + 3) % 4] = *reinterpret_cast<const f32x4 *>(zb + 16 * (q + 3));
            const f32x4 h1n = q + 1 < Q1 ? layer1(q + 1) : h1q;
            const f32x4 *slot = ring + (gstep % NRS) * SLOT, *nslot = ring + ((gstep + 1) % NRS) * SLOT;
#pragma unroll
            for (int n0 = 0; n0 < Q2; n0 += 4) {
                f32x4 wf[4], wn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) wf[u] = wpre[u];
#pragma unroll
                for (int u = 0; u < 4; ++u) wn[u] = (n0 + 4 < Q2) ? slot[(n0 + 4 + u) * 64 + lane] : nslot[u * 64 + lane];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int u = 0; u < 4; ++u) h2[n0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][jj], h1q[jj], h2[n0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) wpre[u] = wn[u];
                __builtin_amdgcn_sched_barrier(0);  // one fragment group ahead, not the whole slice (its reads would take 64-96 registers)
            }
            h1q = h1n;
            ++gstep;
            __syncthreads();
        }
#pragma unroll
        for (int n = 0; n < Q2; ++n) {
            const f32x4 v = h2[n] + *reinterpret_cast<const f32x4 *>(b2l + 16 * n + 4 * g);
            h2[n] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
        }
        // ---- layer 3, transposed (lane = channel 16 n + pt, registers x lane groups = the wave's 16 rows), 16 output chunks at a time
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            f32x4 acc3[16];
#pragma unroll
            for (int n = 0; n < 16; ++n) acc3[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                if (q & 1) step_begin(std::integral_constant<int, 1>{}); else step_begin(std::integral_constant<int, 0>{});
                const f32x4 *slot = ring + (gstep % NRS) * SLOT, *nslot = ring + ((gstep + 1) % NRS) * SLOT;
#pragma unroll
                for (int n0 = 0; n0 < 16; n0 += 4) {
                    f32x4 wf[4], wn[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) wf[u] = wpre[u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) wn[u] = (n0 + 4 < 16) ? slot[(n0 + 4 + u) * 64 + lane] : nslot[u * 64 + lane];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc3[n0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(h2[q][jj], wf[u][jj], acc3[n0 + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) wpre[u] = wn[u];
                    __builtin_amdgcn_sched_barrier(0);
                }
                ++gstep;
                __syncthreads();
            }
            // the wave's maxima over its 16 rows; the pool buffer was read (previous cloud) many barriers ago
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const float m = points16_max_t(acc3[n]);
                if (lane < 16) pool[wave * C3 + half * 256 + 16 * n + pt] = m;
            }
        }
        __syncthreads();
        // max over the eight waves, then bias and ReLU once per channel (max_i relu(x_i + b) = relu(max_i x_i + b))
        {
            float m = pool[tid];
#pragma unroll
            for (int w = 1; w < NWV; ++w) m = fmaxf(m, pool[w * C3 + tid]);
            a.out[(size_t)cloud * a.cout_total + a.cout_off + tid] = fmaxf(m + bias3, 0.f);
        }
        // (the next write to `pool` is a whole layer away: no barrier needed here)
    }
}

template <int C1, int C2, int C3>
int launch_groupall_ring(const SAPreArgs &a, int b, hipStream_t st) {
    constexpr int Q2 = C2 / 16, SLOTF = Q2 > 16 ? Q2 : 16;
    const size_t lds = ((size_t)4 * SLOTF * 64 + C1) * sizeof(f32x4) + (size_t)(8 * C3 + C2) * sizeof(float);
    auto kern = sa_groupall_ring_kernel<C1, C2, C3>;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GP_ELAUNCH;
        done = true;
    }
    hipLaunchKernelGGL(kern, dim3(b < gp_num_cus() ? b : gp_num_cus()), dim3(512), lds, st, a, b);
    return gp_launch_status();
}

template <int C1, int C2, int C3, int NS, bool SPREAD>
int launch_chain_ring(const SAPreArgs &a, int b, hipStream_t st) {
    constexpr int Q1 = C1 / 16, Q2 = (C2 + 15) / 16, Q3 = C3 / 16;
    const size_t lds = ((size_t)(Q1 * Q2 + 3 * Q3) * 64 + C1) * sizeof(f32x4) + (size_t)(16 * Q2 + C3) * sizeof(float);
    if (lds > 160 * 1024) return GP_EINVAL;
    auto kern = sa_chain_ring_kernel<C1, C2, C3, NS, SPREAD>;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GP_ELAUNCH;
        done = true;
    }
    const int ncentres = b * a.np;
    if constexpr (NS > 16) {
        // a small batch (a tracking frame: 5 clouds = 640 neighbourhoods = 80 workgroups): the chunks of a neighbourhood go to different
        // waves while that still fits one workgroup per CU - half the iterations (89 -> ~50 us at 5 clouds)
        constexpr int PT = NS / 16;
        if (ncentres * PT <= 8 * gp_num_cus()) {
            auto kern_s = sa_chain_ring_kernel<C1, C2, C3, NS, SPREAD, true>;
            static bool done_s = false;
            if (!done_s) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern_s), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                    return GP_ELAUNCH;
                done_s = true;
            }
            // (a KERNEL, not hipMemset2DAsync: as a memset node of a graph captured on a side stream - the tracking runner's energy-model
            // graph - the zeroing was not ordered before the kernel on replay, and the atomic max then kept stale values)
            hipLaunchKernelGGL(zero_columns_kernel, dim3((ncentres * (C3 / 4) + 255) / 256), dim3(256), 0, st, a.out + a.cout_off, a.cout_total, C3 / 4,
                               ncentres * (C3 / 4));
            hipLaunchKernelGGL(kern_s, dim3((ncentres * PT + 7) / 8), dim3(512), lds, st, a, ncentres);
            return gp_launch_status();
        }
    }
    int blocks = (ncentres + 7) / 8;
    if (blocks > gp_num_cus()) blocks = gp_num_cus();  // persistent, one 8-wave workgroup per CU (154 KB LDS)
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, a, ncentres);
    return gp_launch_status();
}

template <int C1, int C2, int C3, int NS>
int launch_chain(const SAPreArgs &a, int b, hipStream_t st) {
    const int ncentres = b * a.np;
    int blocks = (ncentres + 3) / 4;
    if (blocks > gp_num_cus() * 8) blocks = gp_num_cus() * 8;  // grid-stride: <= 8 workgroups per CU resident, weights loaded once per wave
    hipLaunchKernelGGL((sa0_chain_kernel<C1, C2, C3, NS>), dim3(blocks), dim3(256), 0, st, a, ncentres);
    return gp_launch_status();
}

template <int P>
int launch(const SAArgs &a, int b, hipStream_t st) {
    const int K0p = gp_round16(a.cin + 3), c1p = gp_round16(a.c1), c2p = gp_round16(a.c2);
    const int lda = (K0p > c2p ? K0p : c2p) + GP_LD_PAD, ldb = c1p + GP_LD_PAD;
    const size_t lds = (size_t)P * (lda + ldb) * sizeof(float);
    if (lds > 160 * 1024) return GP_EINVAL;
    auto kern = sa_mlp_kernel<P>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GP_ELAUNCH;
    }
    const int nrows = a.groupall ? a.n : a.np * a.ns;
    const int tiles = (nrows + P - 1) / P;
    // The GroupAll level of a SMALL batch (a tracking frame: 5 clouds = 20 tiles; one cloud: 4): every tile streams all of both weight
    // matrices (0.77 / 1.15 MB) through one CU while the others idle - 48 us per launch whatever the batch.  While the chip has CUs to spare,
    // 2 or 4 workgroups share a row tile: each recomputes layers 1-2 (a third of the work) and takes a half / quarter of layer 3's channels.
    int split = 1;
    if (a.groupall) {
        const int ncu = gp_num_cus();
        split = tiles * b * 4 <= ncu ? 4 : (tiles * b * 2 <= ncu ? 2 : 1);
    }
    hipLaunchKernelGGL(kern, dim3(tiles, b, split), dim3(256), lds, st, a);
    return gp_launch_status();
}

size_t lds_bytes(int P, const SAArgs &a) {
    const int K0p = gp_round16(a.cin + 3), c1p = gp_round16(a.c1), c2p = gp_round16(a.c2);
    return (size_t)P * ((K0p > c2p ? K0p : c2p) + c1p + 2 * GP_LD_PAD) * sizeof(float);
}

}  // namespace

extern "C" {

int gp_sa_mlp_max(int b, int n, int np, int ns, int cin, int c1, int c2, int c3, const float *xyz, const float *feats_in,
                  const float *new_xyz, const int32_t *idx, const float *wpack1, const float *bias1, const float *wpack2,
                  const float *bias2, const float *wpack3, const float *bias3, float *out, int cout_total, int cout_off,
                  gp_stream_t s) {
    if (b < 0 || n <= 0 || np <= 0 || ns <= 0 || cin < 0 || c1 <= 0 || c2 <= 0 || c3 <= 0) return GP_EINVAL;
    if (!xyz || !wpack1 || !bias1 || !wpack2 || !bias2 || !wpack3 || !bias3 || !out) return GP_EINVAL;
    if ((cin & 3) || (cin > 0 && !feats_in) || (cout_total & 3) || (cout_off & 3) || (c3 & 3)) return GP_EINVAL;
    if (cout_off + c3 > cout_total) return GP_EINVAL;
    const bool groupall = (idx == nullptr);
    if (groupall && (np != 1 || new_xyz != nullptr || ns != n)) return GP_EINVAL;
    if (!groupall && (!new_xyz || ((ns % 16) != 0 && ns != 8))) return GP_EINVAL;
    if (b == 0) return GP_OK;
    SAArgs a{n, np, ns, cin, c1, c2, c3, xyz, feats_in, new_xyz, idx, wpack1, bias1, wpack2, bias2, wpack3, bias3, out, cout_total, cout_off,
             groupall ? 1 : 0};
    hipStream_t st = (hipStream_t)s;
    if (!groupall && ns > 64) return GP_EINVAL;
    // 32-row tiles keep 2-3 workgroups per CU resident (gather of one overlaps the MFMA phase of another); measured
    // faster than 64-row tiles on every level of the light config.  ns = 64 needs 64 rows (one wave owns a neighbourhood).
    // Narrow levels (all widths <= 64: SA level 0) are overhead-bound, not MFMA-bound: 64-row tiles halve the per-tile
    // fixed cost (measured 274 vs 382 us at B = 64).
    const bool narrow = c1 <= 64 && c2 <= 64 && c3 <= 64;
    if (groupall || (ns <= 32 && !narrow)) return launch<32>(a, b, st);
    if (lds_bytes(64, a) <= 150 * 1024) return launch<64>(a, b, st);
    return GP_EINVAL;
}

int gp_point_linear(int rows, int k_in, int n_out, const float *x, const float *wpack, float *z, gp_stream_t s) {
    if (rows < 0 || k_in <= 0 || (k_in & 3) || n_out <= 0 || (n_out & 3) || !x || !wpack || !z) return GP_EINVAL;
    if (rows == 0) return GP_OK;
    // the shapes of the light encoder run with the weights stationary in registers
    // <k-blocks, chunks per wave, 16-row fragments per tile, tile in flight in registers, workgroups per CU>: measured best of
    // the variants that fit the register file (320 clouds: 43 / 93 / 179 us against 76 / 133 / 236 us for the kernel below)
    if (k_in == 96 && n_out == 128) return launch_point_linear_ws<6, 2, 2, true, 4>(rows, n_out, x, wpack, z, (hipStream_t)s);
    if (k_in == 256 && n_out == 256) return launch_point_linear_ws<16, 2, 2, true, 2>(rows, n_out, x, wpack, z, (hipStream_t)s);
    if (k_in == 512 && n_out == 512) return launch_point_linear_ws<32, 1, 2, false, 2>(rows, n_out, x, wpack, z, (hipStream_t)s);
    const size_t lds = (size_t)64 * (gp_round16(k_in) + GP_LD_PAD) * sizeof(float);
    if (lds > 160 * 1024) return GP_EINVAL;
    if (lds > 64 * 1024) {
        static bool done = false;
        if (!done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(point_linear_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess)
                return GP_ELAUNCH;
            done = true;
        }
    }
    hipLaunchKernelGGL(point_linear_kernel, dim3((rows + 63) / 64), dim3(256), lds, (hipStream_t)s, rows, k_in, n_out, x, wpack, z);
    return gp_launch_status();
}

int gp_sa_pre_mlp_max_layout(int hidden_layout, int b, int n, int np, int ns, int c1, int c2, int c3, const float *xyz, const float *new_xyz,
                             const int32_t *idx, const float *z, int zstride, int zoff, const float *wxyz, const float *bias1, const float *wpack2,
                             const float *bias2, const float *wpack3, const float *bias3, float *out, int cout_total, int cout_off, gp_stream_t s) {
    if (hidden_layout != GP_SA_TAIL_PLAIN && hidden_layout != GP_SA_TAIL_SPREAD) return GP_EINVAL;
    if (b < 0 || n <= 0 || np <= 0 || ns <= 0 || c1 <= 0 || c2 <= 0 || c3 <= 0) return GP_EINVAL;
    if (!xyz || !wxyz || !bias1 || !wpack2 || !bias2 || !wpack3 || !bias3 || !out) return GP_EINVAL;
    const bool groupall = !idx && !new_xyz;  // GroupAll level: one neighbourhood = all n points
    if (!groupall && (!idx || !new_xyz)) return GP_EINVAL;
    if (groupall && (np != 1 || ns != n)) return GP_EINVAL;
    if (!groupall && (((ns % 16) != 0 && ns != 8) || ns > 64)) return GP_EINVAL;
    if ((cout_total & 3) || (cout_off & 3) || (c3 & 3) || cout_off + c3 > cout_total) return GP_EINVAL;
    if (z && ((zstride & 3) || (zoff & 3) || zoff + c1 > zstride)) return GP_EINVAL;
    if (b == 0) return GP_OK;
    SAPreArgs a{n, np, ns, c1, c2, c3, zstride, zoff, xyz, new_xyz, z, idx, wxyz, bias1, wpack2, bias2, wpack3, bias3, out, cout_total, cout_off,
                groupall ? 1 : 0};
    if (groupall) {
        // The light encoder's GroupAll shapes: whole clouds on the ring kernel (one cloud per 8-wave workgroup, 82 / 123 us of MFMA time
        // each at the CU's peak - far too coarse to balance a partial round), i.e. full rounds of 256 clouds, or a last round that fills
        // at least three quarters of the chip; the remaining clouds on 32-row tiles (four per cloud, atomic max into the zeroed output).
        // At 320 clouds: 256 + 64 -> 102 + 33 us and 150 + 52 us against 167 and 261 us for tiles alone.
        int nring = 0;
        if (z && n == 128 && c1 == 256 && c3 == 512 && (c2 == 256 || c2 == 384)) {
            const int ncu = gp_num_cus(), full = (b / ncu) * ncu;
            nring = b - full >= (3 * ncu) / 4 ? b : full;
        }
        if (nring > 0) {
            const int rc = c2 == 256 ? launch_groupall_ring<256, 256, 512>(a, nring, (hipStream_t)s) : launch_groupall_ring<256, 384, 512>(a, nring, (hipStream_t)s);
            if (rc != GP_OK || nring == b) return rc;
            a.xyz += (size_t)nring * n * 3;
            a.z += (size_t)nring * n * zstride;
            a.out += (size_t)nring * cout_total;
        }
        return launch_pre<32>(a, b - nring, (hipStream_t)s);  // 64-row tiles: slower where they fit (203 vs 167 us at 320 clouds)
    }
    if (!z) {
        if (c1 == 16 && c2 == 16 && c3 == 32 && ns == 16) return launch_chain<16, 16, 32, 16>(a, b, (hipStream_t)s);
        if (c1 == 32 && c2 == 32 && c3 == 64 && ns == 32) return launch_chain<32, 32, 64, 32>(a, b, (hipStream_t)s);
    }
    if (z && (zoff % 4) == 0 && (zstride % 4) == 0) {
        if (c1 == 64 && c2 == 64 && c3 == 128 && ns == 16) return launch_chain_lds<64, 64, 128, 16>(a, b, (hipStream_t)s);
        if (c1 == 64 && c2 == 96 && c3 == 128 && ns == 32) return launch_chain_lds<64, 96, 128, 32>(a, b, (hipStream_t)s);
        if (c1 == 64 && c2 == 64 && c3 == 128 && ns == 32) return launch_chain_lds<64, 64, 128, 32>(a, b, (hipStream_t)s);  // ClsMSG_CFG_Lighter level 1
        const bool spread = hidden_layout == GP_SA_TAIL_SPREAD;
        if (c1 == 128 && c2 == 196 && c3 == 256 && ns == 16)
            return spread ? launch_chain_ring<128, 196, 256, 16, true>(a, b, (hipStream_t)s) : launch_chain_ring<128, 196, 256, 16, false>(a, b, (hipStream_t)s);
        if (c1 == 128 && c2 == 196 && c3 == 256 && ns == 32)
            return spread ? launch_chain_ring<128, 196, 256, 32, true>(a, b, (hipStream_t)s) : launch_chain_ring<128, 196, 256, 32, false>(a, b, (hipStream_t)s);
    }
    const bool narrow = c1 <= 64 && c2 <= 64 && c3 <= 64;
    if (ns <= 32 && !narrow) return launch_pre<32>(a, b, (hipStream_t)s);
    return launch_pre<64>(a, b, (hipStream_t)s);
}

int gp_sa_pre_mlp_max(int b, int n, int np, int ns, int c1, int c2, int c3, const float *xyz, const float *new_xyz, const int32_t *idx,
                      const float *z, int zstride, int zoff, const float *wxyz, const float *bias1, const float *wpack2, const float *bias2,
                      const float *wpack3, const float *bias3, float *out, int cout_total, int cout_off, gp_stream_t s) {
    return gp_sa_pre_mlp_max_layout(GP_SA_TAIL_PLAIN, b, n, np, ns, c1, c2, c3, xyz, new_xyz, idx, z, zstride, zoff, wxyz, bias1, wpack2, bias2, wpack3,
                                    bias3, out, cout_total, cout_off, s);
}

int gp_sa_tail_position(int c2, int channel) {
    if (c2 <= 0 || channel < 0 || channel >= c2) return GP_EINVAL;
    const int blk = channel / 16, c = channel % 16;
    if (c2 % 16 == 0 || blk != c2 / 16) return channel;  // full blocks keep their order
    return blk * 16 + 4 * (c % 4) + c / 4;
}

int64_t gp_pack_weight_size(int n_out, int k_in) { return (int64_t)(gp_round16(n_out) / 16) * (gp_round16(k_in) / 16) * 256; }

int gp_pack_weight(int n_out, int k_in, const float *W, int ldw, float *packed) {
    if (n_out <= 0 || k_in <= 0 || !W || !packed || ldw < k_in) return GP_EINVAL;
    const int NC = gp_round16(n_out) / 16, KG = gp_round16(k_in) / 16;
    for (int nc = 0; nc < NC; ++nc)
        for (int kg = 0; kg < KG; ++kg)
            for (int lane = 0; lane < 64; ++lane)
                for (int jj = 0; jj < 4; ++jj) {
                    const int n = nc * 16 + (lane & 15), k = kg * 16 + 4 * (lane >> 4) + jj;
                    packed[(((size_t)kg * NC + nc) * 64 + lane) * 4 + jj] = (n < n_out && k < k_in) ? W[(size_t)n * ldw + k] : 0.f;
                }
    return GP_OK;
}

}  // extern "C"
