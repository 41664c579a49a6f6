// Depth + instance mask -> per-instance point clouds on the device: the per-detection body of
// detect_mrcnn_genpose (runners/evaluation_single.py:162-216) - nearest-neighbour crop-and-resize of depth / mask / pixel
// coordinates to img x img (cv2.warpAffine fixed point, utils/datasets_utils.py:82-94), back-projection of the valid
// pixels (depth_to_pcl, :107-118) in raster order, then sample_points (:120-133).  Byte-sized work: one workgroup per
// detection, compaction in raster order of the crop so that the points come out in the reference's order.
#include "gp_common.h"

namespace {

constexpr int AB_BITS = 10;  // OpenCV's fixed-point fraction for warpAffine coordinates

struct RoiArgs {
    int H, W, ninst, img;
    const uint16_t *depth;   // [H,W] millimetres, 0 = no reading
    const uint8_t *masks;    // [H,W,ninst] (Mask-RCNN layout), non-zero = inside the instance
    const double *minv;      // [ninst][6] destination -> source map (the INVERTED affine matrix, row major 2x3)
    float fx, fy, cx, cy;
    float *pcl;              // [ninst][img*img][3] metres, first count[i] rows valid
    int *count;              // [ninst] valid masked pixels
    int *depth_count;        // [ninst] pixels of the crop with a depth reading
};

// One 16-wave workgroup per detection.  Every wave owns a contiguous block of crop rows and walks it twice without any
// workgroup barrier in the loops: pass 1 counts its valid pixels, one barrier publishes the per-wave totals (= each
// wave's base offset in raster order), pass 2 recomputes the pixels and writes them at base + ballot prefix.
constexpr int ROI_WAVES = 16;

struct RoiPixel {
    bool valid, has_depth;
    int sx, sy;
    float d;
};

__device__ __forceinline__ RoiPixel roi_pixel(const RoiArgs &a, int inst, long long X0, long long Y0, double m00, double m10, int x) {
    RoiPixel p{false, false, 0, 0, 0.f};
    if (x >= a.img) return p;
    // adelta[x] = cvRound(M00 * x * 1024), source = (X0 + adelta[x]) >> 10 (arithmetic shift), saturate_cast<short>
    long long sxl = (X0 + llrint(m00 * x * (double)(1 << AB_BITS))) >> AB_BITS;
    long long syl = (Y0 + llrint(m10 * x * (double)(1 << AB_BITS))) >> AB_BITS;
    sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);
    syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
    if (sxl >= 0 && sxl < a.W && syl >= 0 && syl < a.H) {
        p.sx = (int)sxl, p.sy = (int)syl;
        const size_t q = (size_t)p.sy * a.W + p.sx;
        const uint16_t dv = a.depth[q];
        p.has_depth = dv > 0;
        p.d = (float)dv;
        p.valid = p.has_depth && a.masks[q * a.ninst + inst] != 0;
    }
    return p;
}

__global__ __launch_bounds__(64 * ROI_WAVES) void roi_cloud_kernel(RoiArgs a) {
    __shared__ int wave_tot[ROI_WAVES], wave_dep[ROI_WAVES];
    const int inst = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *M = a.minv + (size_t)inst * 6;
    const double m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[3], m11 = M[4], m12 = M[5];
    float *out = a.pcl + (size_t)inst * a.img * a.img * 3;
    const int rpw = (a.img + ROI_WAVES - 1) / ROI_WAVES;
    const int r0 = wave * rpw, r1 = (r0 + rpw < a.img) ? r0 + rpw : a.img;
    int cnt = 0, dep = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int off = 0;
        if (pass == 1) {
            if (lane == 0) wave_tot[wave] = cnt, wave_dep[wave] = dep;
            __syncthreads();
            for (int w = 0; w < wave; ++w) off += wave_tot[w];
        }
        for (int y = r0; y < r1; ++y) {
            // X0 = cvRound((M01*y + M02) * 1024) + 512   (rint = round half to even = cvRound)
            const long long X0 = llrint((m01 * y + m02) * (double)(1 << AB_BITS)) + (1 << (AB_BITS - 1));
            const long long Y0 = llrint((m11 * y + m12) * (double)(1 << AB_BITS)) + (1 << (AB_BITS - 1));
            for (int x0 = 0; x0 < a.img; x0 += 64) {
                const RoiPixel p = roi_pixel(a, inst, X0, Y0, m00, m10, x0 + lane);
                const unsigned long long bal = __ballot(p.valid);
                if (pass == 0) {
                    cnt += __popcll(bal);
                    dep += __popcll(__ballot(p.has_depth));
                } else {
                    if (p.valid) {
                        // depth_to_pcl, all float32: (x - cx) * d / fx, (y - cy) * d / fy, d; then / 1000
                        float *o = out + (size_t)(off + __popcll(bal & ((1ull << lane) - 1))) * 3;
                        o[0] = (((float)p.sx - a.cx) * p.d / a.fx) / 1000.0f;
                        o[1] = (((float)p.sy - a.cy) * p.d / a.fy) / 1000.0f;
                        o[2] = p.d / 1000.0f;
                    }
                    off += __popcll(bal);
                }
            }
        }
    }
    if (threadIdx.x == 0) {
        int c = 0, d = 0;
        for (int w = 0; w < ROI_WAVES; ++w) c += wave_tot[w], d += wave_dep[w];
        a.count[inst] = c, a.depth_count[inst] = d;
    }
}

// sample_points: count < n -> tile (row k % count); count > n -> rows ids[k] (first n of a host permutation); else copy
__global__ void cloud_sample_kernel(int cap, int npts, const float *__restrict__ pcl, const int *__restrict__ count,
                                    const int32_t *__restrict__ ids, float *__restrict__ out) {
    const int inst = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npts) return;
    const int c = count[inst];
    if (c <= 0) return;
    int src = c > npts ? (ids ? ids[(size_t)inst * npts + k] : k) : k % c;
    src = src < 0 ? 0 : (src >= c ? c - 1 : src);
    const float *s = pcl + ((size_t)inst * cap + src) * 3;
    float *o = out + ((size_t)inst * npts + k) * 3;
    o[0] = s[0], o[1] = s[1], o[2] = s[2];
}

}  // namespace

extern "C" {

int gp_roi_to_cloud(int h, int w, int ninst, int img, const uint16_t *depth, const uint8_t *masks, const double *minv, float fx, float fy,
                    float cx, float cy, float *pcl, int32_t *count, int32_t *depth_count, gp_stream_t s) {
    if (h <= 0 || w <= 0 || ninst < 0 || img <= 0 || !depth || !masks || !minv || !pcl || !count || !depth_count) return GP_EINVAL;
    if (h > 32767 || w > 32767) return GP_EINVAL;
    if (ninst == 0) return GP_OK;
    RoiArgs a{h, w, ninst, img, depth, masks, minv, fx, fy, cx, cy, pcl, count, depth_count};
    hipLaunchKernelGGL(roi_cloud_kernel, dim3(ninst), dim3(64 * ROI_WAVES), 0, (hipStream_t)s, a);
    return gp_launch_status();
}

int gp_cloud_sample(int ninst, int cap, int npts, const float *pcl, const int32_t *count, const int32_t *ids, float *out, gp_stream_t s) {
    if (ninst < 0 || cap <= 0 || npts <= 0 || !pcl || !count || !out) return GP_EINVAL;
    if (ninst == 0) return GP_OK;
    hipLaunchKernelGGL(cloud_sample_kernel, dim3((npts + 255) / 256, ninst), dim3(256), 0, (hipStream_t)s, cap, npts, pcl, count, ids, out);
    return gp_launch_status();
}

}  // extern "C"
