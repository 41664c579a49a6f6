/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path
 * (genpose_amd/*); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 *
 * Plain-C restatement of the forward kernels of the reference's CUDA extension
 * `pointnet2_cuda` (reference: networks/pts_encoder/pointnet2_utils/pointnet2/src/).
 * The reference kernels are CUDA-only and cannot be built or run in this image
 * (no nvcc, no NVIDIA device) -> at the .cu level this oracle is
 * "PARITY UNPINNED": it follows the .cu text line by line, but no golden
 * vector produced by the CUDA code exists to check it against.
 *
 * Arithmetic convention (`arith`, first argument of every function that evaluates a sum of three
 * products).  The .cu text says `dx*dx + dy*dy + dz*dz` (sampling_gpu.cu:133, ball_query_gpu.cu:33,
 * interpolate_gpu.cu:36) and `w0*p0 + w1*p1 + w2*p2` (interpolate_gpu.cu:95); the reference builds with
 * plain `nvcc -O2` (setup.py:19-20), i.e. --fmad=true, so nvcc contracts - HOW is a property of its
 * compiler, which this image does not have.  Three conventions, all restated here (DESIGN.md §5 has the
 * evidence for the default):
 *   GPO_ARITH_A = 0   fma(c,c, fma(b,b, a*a))    first product rounded, then fused left to right
 *   GPO_ARITH_B = 1   fma(c,c, fma(a,a, b*b))    what LLVM's DAG combiner and GCC emit for this text
 *                                                (the first fadd fuses its LEFT operand's multiply): DEFAULT
 *   GPO_ARITH_C = 2   (a*a + b*b) + c*c          no contraction (nvcc --fmad=false)
 * Build with -ffp-contract=off so the compiler adds no contraction of its own.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { GPO_ARITH_A = 0, GPO_ARITH_B = 1, GPO_ARITH_C = 2 };

/* a0*b0 + a1*b1 + a2*b2 under the three conventions */
static inline float dot3(int arith, float a0, float b0, float a1, float b1, float a2, float b2) {
    if (arith == GPO_ARITH_A) {
        float m = a0 * b0;
        return fmaf(a2, b2, fmaf(a1, b1, m));
    }
    if (arith == GPO_ARITH_B) {
        float m = a1 * b1;
        return fmaf(a2, b2, fmaf(a0, b0, m));
    }
    float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
    float s = p0 + p1;
    return s + p2;
}

static inline float sqdist(int arith, float ax, float ay, float az, float bx, float by, float bz) {
    /* operand order as written in the .cu: (a - b) */
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return dot3(arith, dx, dx, dy, dy, dz, dz);
}
int gpo_arith_valid(int arith) { return arith >= GPO_ARITH_A && arith <= GPO_ARITH_C; }
/* one squared distance (tests: the three conventions differ on crafted inputs) */
float gpo_sqdist(int arith, const float *a, const float *b) { return sqdist(arith, a[0], a[1], a[2], b[0], b[1], b[2]); }

/* cuda_utils.h:10-14  opt_n_threads */
static int opt_n_threads(int work_size) {
    int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}
int gpo_opt_n_threads(int n) { return opt_n_threads(n); }

/*
 * sampling_gpu.cu:86-209 furthest_point_sampling_kernel<block_size>.
 * The shared-memory tree is simulated literally (slot arrays dists/dists_i,
 * strides S/2 ... 1, `v2 > v1 ? i2 : i1`), so the tie rule is whatever the
 * tree produces - not an independently derived closed form.
 * dataset (B,N,3)  temp (B,N) in/out  idxs (B,M) out.
 */
int gpo_furthest_point_sampling(int arith, int b, int n, int m, const float *dataset, float *temp, int32_t *idxs) {
    if (!gpo_arith_valid(arith)) return -1;
    if (m <= 0) return 0;
    const int S = opt_n_threads(n);
#pragma omp parallel for schedule(dynamic)
    for (int bi = 0; bi < b; ++bi) {
        float *dists = (float *)malloc(sizeof(float) * S);
        int *dists_i = (int *)malloc(sizeof(int) * S);
        const float *ds = dataset + (size_t)bi * n * 3;
        float *tp = temp + (size_t)bi * n;
        int32_t *out = idxs + (size_t)bi * m;
        int old = 0;
        out[0] = old;
        for (int j = 1; j < m; ++j) {
            float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
            for (int tid = 0; tid < S; ++tid) {
                int besti = 0;
                float best = -1.0f;
                for (int k = tid; k < n; k += S) {
                    float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
                    float d = sqdist(arith, x2, y2, z2, x1, y1, z1);
                    float d2 = fminf(d, tp[k]);
                    tp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int s = S / 2; s >= 1; s >>= 1) {
                for (int tid = 0; tid < s; ++tid) { /* __update(dists, dists_i, tid, tid+s) */
                    float v1 = dists[tid], v2 = dists[tid + s];
                    int i1 = dists_i[tid], i2 = dists_i[tid + s];
                    dists[tid] = v1 > v2 ? v1 : v2; /* max(v1,v2) */
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            }
            old = dists_i[0];
            out[j] = old;
        }
        free(dists);
        free(dists_i);
    }
    return 1;
}

/* sampling_gpu.cu:8-24 gather_points_kernel_fast: points (B,C,N), idx (B,M) -> out (B,C,M) */
int gpo_gather_points(int b, int c, int n, int m, const float *points, const int32_t *idx, float *out) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int p = 0; p < m; ++p)
                out[((size_t)bi * c + ci) * m + p] = points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + p]];
    return 1;
}

/* ball_query_gpu.cu:9-45: new_xyz (B,M,3), xyz (B,N,3) -> idx (B,M,nsample); idx is NOT cleared here
 * (the Python caller pre-zeroes it, pointnet2_utils.py:219). */
int gpo_ball_query(int arith, int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int32_t *idx) {
    if (!gpo_arith_valid(arith)) return -1;
    float radius2 = radius * radius;
#pragma omp parallel for schedule(dynamic)
    for (int bi = 0; bi < b; ++bi) {
        const float *X = xyz + (size_t)bi * n * 3;
        for (int p = 0; p < m; ++p) {
            const float *c = new_xyz + ((size_t)bi * m + p) * 3;
            int32_t *o = idx + ((size_t)bi * m + p) * nsample;
            float nx = c[0], ny = c[1], nz = c[2];
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                float d2 = sqdist(arith, nx, ny, nz, X[k * 3 + 0], X[k * 3 + 1], X[k * 3 + 2]);
                if (d2 < radius2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) o[l] = k;
                    o[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
    }
    return 1;
}

/* group_points_gpu.cu:47-66: points (B,C,N), idx (B,np,ns) -> out (B,C,np,ns) */
int gpo_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int32_t *idx, float *out) {
#pragma omp parallel for schedule(dynamic)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *src = points + ((size_t)bi * c + ci) * n;
            float *dst = out + ((size_t)bi * c + ci) * npoints * nsample;
            const int32_t *id = idx + (size_t)bi * npoints * nsample;
            for (int q = 0; q < npoints * nsample; ++q) dst[q] = src[id[q]];
        }
    return 1;
}

/* interpolate_gpu.cu:9-52 three_nn_kernel_fast: unknown (B,N,3), known (B,M,3) -> dist2 (B,N,3), idx (B,N,3).
 * best1..3 are doubles initialised to 1e40, d is float compared after promotion. */
int gpo_three_nn(int arith, int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx) {
    for (int bi = 0; bi < b; ++bi)
        for (int p = 0; p < n; ++p) {
            const float *u = unknown + ((size_t)bi * n + p) * 3;
            const float *K = known + (size_t)bi * m * 3;
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                float d = sqdist(arith, u[0], u[1], u[2], K[k * 3 + 0], K[k * 3 + 1], K[k * 3 + 2]);
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            float *o = dist2 + ((size_t)bi * n + p) * 3;
            int32_t *oi = idx + ((size_t)bi * n + p) * 3;
            o[0] = (float)best1; o[1] = (float)best2; o[2] = (float)best3;
            oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
        }
    return 1;
}

/* interpolate_gpu.cu:77-96 three_interpolate_kernel_fast: points (B,C,M), idx/weight (B,N,3) -> out (B,C,N).
 * `w0*p0 + w1*p1 + w2*p2`: the same three conventions as the squared distance (dot3). */
int gpo_three_interpolate(int arith, int b, int c, int m, int n, const float *points, const int32_t *idx, const float *weight, float *out) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *src = points + ((size_t)bi * c + ci) * m;
            for (int p = 0; p < n; ++p) {
                const float *w = weight + ((size_t)bi * n + p) * 3;
                const int32_t *id = idx + ((size_t)bi * n + p) * 3;
                out[((size_t)bi * c + ci) * n + p] = dot3(arith, w[0], src[id[0]], w[1], src[id[1]], w[2], src[id[2]]);
            }
        }
    return 1;
}

/* backward kernels (group_points_gpu.cu:8-25, sampling_gpu.cu:46-63, interpolate_gpu.cu:120-142):
 * atomicAdd scatters; sequential accumulation here (float add order differs from the GPU's
 * nondeterministic atomic order -> tolerance-level comparison only). */
int gpo_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int32_t *idx, float *grad_points) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int q = 0; q < npoints * nsample; ++q)
                grad_points[((size_t)bi * c + ci) * n + idx[(size_t)bi * npoints * nsample + q]] +=
                    grad_out[((size_t)bi * c + ci) * npoints * nsample + q];
    return 1;
}
int gpo_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx, float *grad_points) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int p = 0; p < m; ++p)
                grad_points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + p]] += grad_out[((size_t)bi * c + ci) * m + p];
    return 1;
}
int gpo_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx, const float *weight, float *grad_points) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int p = 0; p < n; ++p) {
                float g = grad_out[((size_t)bi * c + ci) * n + p];
                const float *w = weight + ((size_t)bi * n + p) * 3;
                const int32_t *id = idx + ((size_t)bi * n + p) * 3;
                float *gp = grad_points + ((size_t)bi * c + ci) * m;
                gp[id[0]] += g * w[0];
                gp[id[1]] += g * w[1];
                gp[id[2]] += g * w[2];
            }
    return 1;
}
