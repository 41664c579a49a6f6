// Energy ranking and top-k pose aggregation, one wave per cloud.
// Reference: sort_poses_by_energy (networks/reward.py:131-155: rotation and translation are ranked INDEPENDENTLY by
// their own energy channel), sort_sRT_by_energy(..., ratio, 'average') (utils/sgpa_utils.py:897-954) ==
// cal_average_sRT (runners/evaluation_tracking.py:60-77): 6-D -> matrix (Gram-Schmidt) -> quaternion (pytorch3d
// matrix_to_quaternion) -> sign-align w>0 -> A = mean(q q^T) -> eigenvector of the largest eigenvalue
// (utils/misc.py:227-249) + mean translation.  The reference does this through .cpu().numpy().tolist() index lists
// and torch.linalg.eigh; here it is one launch (4x4 Jacobi in f64 registers).
#include "gp_common.h"

namespace {

template <typename T>
__device__ __forceinline__ void rot6_to_matrix(const T *p, double R[3][3]) {
    // get_rot_matrix('rot_matrix') (utils/misc.py:136): columns b1, b2, b3 = b1 x b2
    double a1[3] = {(double)p[0], (double)p[1], (double)p[2]}, a2[3] = {(double)p[3], (double)p[4], (double)p[5]};
    double n1 = sqrt(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    n1 = n1 > 1e-12 ? n1 : 1e-12;
    double b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    double d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    double c[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    double n2 = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    n2 = n2 > 1e-12 ? n2 : 1e-12;
    double b2[3] = {c[0] / n2, c[1] / n2, c[2] / n2};
    double b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    for (int i = 0; i < 3; ++i) R[i][0] = b1[i], R[i][1] = b2[i], R[i][2] = b3[i];
}

__device__ __forceinline__ void matrix_to_quat(const double m[3][3], double q[4]) {
    // pytorch3d v0.7.2 matrix_to_quaternion: candidate with the largest |q_i|, divisor floored at 0.1
    // (every array index below is a compile-time constant: the arrays live in registers, no scratch)
    double qa[4] = {1.0 + m[0][0] + m[1][1] + m[2][2], 1.0 + m[0][0] - m[1][1] - m[2][2], 1.0 - m[0][0] + m[1][1] - m[2][2],
                    1.0 - m[0][0] - m[1][1] + m[2][2]};
#pragma unroll
    for (int i = 0; i < 4; ++i) qa[i] = qa[i] > 0 ? sqrt(qa[i]) : 0.0;
    int best = 0;
    double qb = qa[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (qa[i] > qb) best = i, qb = qa[i];
    double c[4];
    if (best == 0) {
        c[0] = qa[0] * qa[0], c[1] = m[2][1] - m[1][2], c[2] = m[0][2] - m[2][0], c[3] = m[1][0] - m[0][1];
    } else if (best == 1) {
        c[0] = m[2][1] - m[1][2], c[1] = qa[1] * qa[1], c[2] = m[1][0] + m[0][1], c[3] = m[0][2] + m[2][0];
    } else if (best == 2) {
        c[0] = m[0][2] - m[2][0], c[1] = m[1][0] + m[0][1], c[2] = qa[2] * qa[2], c[3] = m[1][2] + m[2][1];
    } else {
        c[0] = m[1][0] - m[0][1], c[1] = m[2][0] + m[0][2], c[2] = m[2][1] + m[1][2], c[3] = qa[3] * qa[3];
    }
    const double den = 2.0 * (qb > 0.1 ? qb : 0.1);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = c[i] / den;
}

// eigenvector of the largest eigenvalue of a symmetric 4x4 (cyclic Jacobi).  The (p, q) sweeps are unrolled: A and V are indexed by
// constants only and stay in registers (round 5's dynamically indexed form cost 64 scratch instructions on the tracking path).
__device__ __forceinline__ void top_eigvec4(double (&A)[4][4], double (&v)[4]) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        if (off < 1e-60) break;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double apq = A[p][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    double bv = A[0][0];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = V[i][0];
#pragma unroll
    for (int b = 1; b < 4; ++b)
        if (A[b][b] > bv) {
            bv = A[b][b];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = V[i][b];
        }
}

// [7] (w, x, y, z, t) f32 -> [4][4] f32: pytorch3d quaternion_to_matrix (two_s = 2 / |q|^2) + the translation
__device__ __forceinline__ void quat_trans_to_rt(const float *q, float *o) {
    const float r = q[0], a = q[1], b = q[2], c = q[3];
    const float two_s = 2.0f / (((r * r + a * a) + b * b) + c * c);
    o[0] = 1.f - two_s * (b * b + c * c), o[1] = two_s * (a * b - c * r), o[2] = two_s * (a * c + b * r), o[3] = q[4];
    o[4] = two_s * (a * b + c * r), o[5] = 1.f - two_s * (a * a + c * c), o[6] = two_s * (b * c - a * r), o[7] = q[5];
    o[8] = two_s * (a * c - b * r), o[9] = two_s * (b * c + a * r), o[10] = 1.f - two_s * (a * a + b * b), o[11] = q[6];
    o[12] = o[13] = o[14] = 0.f, o[15] = 1.f;
}

// One wave per cloud: ranking, sorted copies, aggregation - and (optional) the 4x4 forms the runners hand on: sorted_rt [k][4][4] f64 =
// gp_pose9_to_rt of the sorted poses, avg_rt [4][4] f32 = gp_quat_trans_to_rt of avg_pose (same arithmetic, same bits), so that the ranking
// step of a tracking frame is this ONE launch.
template <typename T>
__global__ __launch_bounds__(64) void rank_aggregate_kernel(int k, int sel, const T *__restrict__ poses, const float *__restrict__ energy,
                                                            T *__restrict__ sorted_poses, float *__restrict__ sorted_energy,
                                                            int32_t *__restrict__ order, float *__restrict__ avg_pose,
                                                            double *__restrict__ sorted_rt, float *__restrict__ avg_rt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *e = reinterpret_cast<float *>(smem);                   // [k][2]
    int *ord = reinterpret_cast<int *>(e + 2 * k);                // [k][2]: candidate index at each rank
    double *quat = reinterpret_cast<double *>(ord + 2 * k);  // [sel][4] (offset 16k bytes: 8-byte aligned)
    const int b = blockIdx.x, tid = threadIdx.x;
    poses += (size_t)b * k * 9;
    energy += (size_t)b * k * 2;
    for (int i = tid; i < 2 * k; i += 64) e[i] = energy[i];
    __syncthreads();
    // stable descending rank by counting (torch.sort(descending=True); ties keep candidate order)
    for (int i = tid; i < k; i += 64) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float ei = e[2 * i + c];
            int r = 0;
            for (int j = 0; j < k; ++j) {
                const float ej = e[2 * j + c];
                r += (ej > ei) || (ej == ei && j < i);
            }
            ord[2 * r + c] = i;
        }
    }
    __syncthreads();
    for (int r = tid; r < k; r += 64) {
        const int ir = ord[2 * r + 0], it = ord[2 * r + 1];
        const size_t o = ((size_t)b * k + r);
        if (order) {
            order[o * 2 + 0] = ir;
            order[o * 2 + 1] = it;
        }
        if (sorted_energy) {
            sorted_energy[o * 2 + 0] = e[2 * ir + 0];
            sorted_energy[o * 2 + 1] = e[2 * it + 1];
        }
        T sp[9];
#pragma unroll
        for (int j = 0; j < 6; ++j) sp[j] = poses[(size_t)ir * 9 + j];
#pragma unroll
        for (int j = 6; j < 9; ++j) sp[j] = poses[(size_t)it * 9 + j];
        if (sorted_poses) {
#pragma unroll
            for (int j = 0; j < 9; ++j) sorted_poses[o * 9 + j] = sp[j];
        }
        if (sorted_rt) {
            double R[3][3];
            rot6_to_matrix<T>(sp, R);
            double *m = sorted_rt + o * 16;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                m[4 * rr + 0] = R[rr][0], m[4 * rr + 1] = R[rr][1], m[4 * rr + 2] = R[rr][2];
                m[4 * rr + 3] = (double)sp[6 + rr];
            }
            m[12] = m[13] = m[14] = 0.0, m[15] = 1.0;
        }
    }
    if (!avg_pose) return;
    for (int r = tid; r < sel; r += 64) {
        double R[3][3], q[4];
        rot6_to_matrix<T>(poses + (size_t)ord[2 * r + 0] * 9, R);
        matrix_to_quat(R, q);
        const double sgn = q[0] > 0 ? 1.0 : -1.0;  // ((q_w > 0) - 0.5) * 2  (misc.py:242)
#pragma unroll
        for (int i = 0; i < 4; ++i) quat[4 * r + i] = sgn * q[i];
    }
    __syncthreads();
    // A = mean(q q^T): lane l < 16 sums element (l / 4, l % 4) over the selected candidates in rank order (the order the serial loop of
    // rounds 1-5 used: same bits); every lane then holds all of A (wave shuffles) and runs the Jacobi iteration redundantly
    double mine = 0.0;
    {
        const int i = (tid >> 2) & 3, j = tid & 3;
        for (int r = 0; r < sel; ++r) mine += quat[4 * r + i] * quat[4 * r + j];
        mine /= (double)sel;
    }
    double A[4][4], v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) A[i][j] = __shfl(mine, 4 * i + j, 64);
    top_eigvec4(A, v);
    const double sgn = v[0] > 0 ? 1.0 : -1.0;
    // mean translation: lanes 0..2, one component each, candidates in rank order of the translation energy
    double tm = 0.0;
    if (tid < 3) {
        for (int r = 0; r < sel; ++r) tm += (double)poses[(size_t)ord[2 * r + 1] * 9 + 6 + tid];
        tm /= (double)sel;
    }
    float qt[7];
#pragma unroll
    for (int i = 0; i < 4; ++i) qt[i] = (float)(sgn * v[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) qt[4 + i] = (float)__shfl(tm, i, 64);
    if (tid == 0) {
        float *o = avg_pose + (size_t)b * 7;
#pragma unroll
        for (int i = 0; i < 7; ++i) o[i] = qt[i];
        if (avg_rt) quat_trans_to_rt(qt, avg_rt + (size_t)b * 16);
    }
}

}  // namespace

// [n][9] poses (two rotation columns + translation) -> [n][4][4] f64 homogeneous matrices (evaluation_single.py:325-332: get_rot_matrix
// on float64 rows, translation in the last column)
template <typename T>
__global__ void pose9_to_rt_kernel(int n, const T *__restrict__ pose, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double R[3][3];
    rot6_to_matrix<T>(pose + (size_t)i * 9, R);
    double *o = out + (size_t)i * 16;
    for (int r = 0; r < 3; ++r) {
        o[4 * r + 0] = R[r][0], o[4 * r + 1] = R[r][1], o[4 * r + 2] = R[r][2];
        o[4 * r + 3] = (double)pose[(size_t)i * 9 + 6 + r];
    }
    o[12] = o[13] = o[14] = 0.0, o[15] = 1.0;
}

// [n][7] (w, x, y, z, t) f32 -> [n][4][4] f32 (quat_trans_to_rt above)
__global__ void quat_trans_to_rt_kernel(int n, const float *__restrict__ qt, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    quat_trans_to_rt(qt + (size_t)i * 7, out + (size_t)i * 16);
}

extern "C" {

int gp_pose9_to_rt(int n, int is_f64, const void *pose, double *out, gp_stream_t s) {
    if (n < 0 || !pose || !out) return GP_EINVAL;
    if (n == 0) return GP_OK;
    if (is_f64)
        hipLaunchKernelGGL(pose9_to_rt_kernel<double>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, n, (const double *)pose, out);
    else
        hipLaunchKernelGGL(pose9_to_rt_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, n, (const float *)pose, out);
    return gp_launch_status();
}

int gp_quat_trans_to_rt(int n, const float *quat_trans, float *out, gp_stream_t s) {
    if (n < 0 || !quat_trans || !out) return GP_EINVAL;
    if (n == 0) return GP_OK;
    hipLaunchKernelGGL(quat_trans_to_rt_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, n, quat_trans, out);
    return gp_launch_status();
}

int gp_rank_aggregate_rt(int b, int k, int sel, int is_f64, const void *poses, const float *energy, void *sorted_poses, float *sorted_energy,
                         int32_t *order, float *avg_pose, double *sorted_rt, float *avg_rt, gp_stream_t s) {
    if (b < 0 || k <= 0 || k > 2048 || !poses || !energy) return GP_EINVAL;
    if (avg_pose && (sel <= 0 || sel > k)) return GP_EINVAL;
    if (avg_rt && !avg_pose) return GP_EINVAL;
    if (b == 0) return GP_OK;
    const size_t lds = (size_t)k * 16 + 16 + (size_t)(avg_pose ? sel : 0) * 32;
    if (is_f64)
        hipLaunchKernelGGL(rank_aggregate_kernel<double>, dim3(b), dim3(64), lds, (hipStream_t)s, k, sel, (const double *)poses, energy,
                           (double *)sorted_poses, sorted_energy, order, avg_pose, sorted_rt, avg_rt);
    else
        hipLaunchKernelGGL(rank_aggregate_kernel<float>, dim3(b), dim3(64), lds, (hipStream_t)s, k, sel, (const float *)poses, energy,
                           (float *)sorted_poses, sorted_energy, order, avg_pose, sorted_rt, avg_rt);
    return gp_launch_status();
}

int gp_rank_aggregate(int b, int k, int sel, int is_f64, const void *poses, const float *energy, void *sorted_poses, float *sorted_energy,
                      int32_t *order, float *avg_pose, gp_stream_t s) {
    return gp_rank_aggregate_rt(b, k, sel, is_f64, poses, energy, sorted_poses, sorted_energy, order, avg_pose, nullptr, nullptr, s);
}

}  // extern "C"
