/*
 * genpose_hip.h - C ABI of libgenpose_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for GenPose's inference hot path.  Every entry point
 *   - takes raw DEVICE pointers, plain ints/floats and a hipStream_t passed as void* (0 = null stream),
 *   - is stateless, allocation-free, never synchronises and never reads results back on the host
 *     (safe to capture in a hipGraph),
 *   - returns 0 on success or a negative GP_E* code (never exit()s, unlike the reference launchers,
 *     e.g. ball_query_gpu.cu:62-66).
 * Caller owns every buffer, outputs included (same convention as the reference's pybind wrappers,
 * pointnet2_utils.py:26-27,56,95-96,129,173,219).
 *
 * Section A replaces, one for one, the nine pybind functions of the reference's CUDA extension
 * `pointnet2_cuda` (networks/pts_encoder/pointnet2_utils/pointnet2/src/pointnet2_api.cpp:10-24).
 * Sections B-D are the fused MI355X-native entry points that replace the Python/ATen glue above them
 * (pointnet2_modules.py:19-56, scorenet.py:178-222, samplers.py:102-227, reward.py:131-155,
 * sgpa_utils.py:897-954).
 */
#ifndef GENPOSE_HIP_H
#define GENPOSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GP_OK 0
#define GP_EINVAL (-1)   /* bad size / null pointer / unsupported shape */
#define GP_ELAUNCH (-2)  /* hipGetLastError() != hipSuccess after a launch */
#define GP_EARCH (-3)    /* device is not gfx950 */

typedef void *gp_stream_t; /* hipStream_t */

/* ---------------------------------------------------------------------------------------------------------------
 * Arithmetic convention of the reference's three-product sums.  The reference's kernels say
 *     d = dx*dx + dy*dy + dz*dz          sampling_gpu.cu:133, ball_query_gpu.cu:33, interpolate_gpu.cu:36
 *     o = w0*p0 + w1*p1 + w2*p2          interpolate_gpu.cu:95
 * and are built with plain `nvcc -O2` (setup.py:19-20: --fmad=true), so nvcc decides how they contract - and which product
 * is fused decides, for distances that agree to an ulp, WHICH index furthest point sampling or a ball query returns.  There is
 * no CUDA toolchain here to settle it, so the convention is an explicit, tested parameter (compile-time inside every kernel):
 *     GP_ARITH_A   fma(c,c, fma(b,b, a*a))   the first product rounded on its own, the other two fused left to right
 *     GP_ARITH_B   fma(c,c, fma(a,a, b*b))   what LLVM's DAG combiner and GCC's widening-mul pass emit for this text: the
 *                                            inner fadd fuses its LEFT operand's multiply, the outer one the remaining product
 *     GP_ARITH_C   (a*a + b*b) + c*c         no contraction (nvcc --fmad=false)
 * GP_ARITH_DEFAULT is what every entry point WITHOUT an `_arith` suffix uses (DESIGN.md section 5 has the evidence and the
 * measured index differences between the conventions); the `_arith` twin of an entry point takes the convention first. */
#define GP_ARITH_A 0
#define GP_ARITH_B 1
#define GP_ARITH_C 2
#define GP_ARITH_DEFAULT GP_ARITH_B
int gp_arith_default(void); /* the GP_ARITH_DEFAULT this library was built with */

/* ---------------------------------------------------------------------------------------------------------------
 * E. Depth + instance mask -> point clouds (the step right before the path).
 * Replaces the per-detection body of detect_mrcnn_genpose (runners/evaluation_single.py:162-216):
 *   crop_resize_by_warp_affine(INTER_NEAREST) of raw depth, mask & (depth > 0) and the pixel-coordinate map
 *   (utils/datasets_utils.py:82-94; OpenCV's 10-bit fixed-point nearest sampling), depth_to_pcl (:107-118, float32) on the
 *   pixels with depth > 0 inside the mask, in raster order of the img x img crop, divided by 1000 (metres).
 *   depth [h,w] uint16 mm; masks [h,w,ninst] uint8 (Mask-RCNN layout); minv [ninst][6] = the INVERSE of the 2x3 matrix that
 *   get_affine_transform (:96-136) builds from get_bbox's window (doubles, row major): source = minv . (x, y, 1);
 *   pcl [ninst][img*img][3] f32 (first count[i] rows written); count / depth_count [ninst] = number of valid masked pixels /
 *   of crop pixels with a depth reading (the reference skips an instance when either is <= 1, :201-208).
 * gp_cloud_sample = sample_points (:120-133): out[i][k] = pcl[i][k % count] when count <= npts (tiling), else pcl[i][ids[i][k]]
 *   (ids = first npts entries of a permutation of count, drawn by the caller; NULL -> the first npts rows). */
int gp_roi_to_cloud(int h, int w, int ninst, int img, const uint16_t *depth, const uint8_t *masks, const double *minv, float fx, float fy,
                    float cx, float cy, float *pcl, int32_t *count, int32_t *depth_count, gp_stream_t s);
int gp_cloud_sample(int ninst, int cap, int npts, const float *pcl, const int32_t *count, const int32_t *ids, float *out, gp_stream_t s);

/* Library / device identification: returns ABI version; writes gcnArchName of the current device. */
int gp_version(void);
int gp_device_arch(char *buf, int buflen);

/* ------------------------------------------------------------------------------------------------
 * A. pointnet2_cuda operator API (reference: pointnet2_api.cpp:10-24)
 * ------------------------------------------------------------------------------------------------ */

/* furthest_point_sampling_wrapper (sampling.cpp:40-51, sampling_gpu.cu:86-253).
 * xyz [b,n,3] f32; temp [b,n] f32 in/out (caller fills 1e10, pointnet2_utils.py:27; must be >= 0);
 * idx [b,m] i32 out.  idx[:,0] = 0; ties resolved exactly as the reference's shared-memory tree does. */
int gp_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int32_t *idx, gp_stream_t s);
int gp_furthest_point_sampling_arith(int arith, int b, int n, int m, const float *xyz, float *temp, int32_t *idx, gp_stream_t s);

/* gather_points_wrapper (sampling.cpp:13-23, sampling_gpu.cu:8-44): points [b,c,n], idx [b,m] -> out [b,c,m]. */
int gp_gather_points(int b, int c, int n, int m, const float *points, const int32_t *idx, float *out, gp_stream_t s);
/* gather_points_grad_wrapper (sampling_gpu.cu:46-83): grad_out [b,c,m], idx [b,m] -> grad_points [b,c,n] (+=, atomic). */
int gp_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx, float *grad_points, gp_stream_t s);

/* ball_query_wrapper (ball_query.cpp:16-27, ball_query_gpu.cu:9-67).
 * new_xyz [b,m,3], xyz [b,n,3] -> idx [b,m,nsample] i32.  First `nsample` points in index order with
 * d2 < radius^2 (strict, radius^2 in f32); the first hit pre-fills every slot; rows with no hit are
 * left untouched (caller pre-zeroes idx, pointnet2_utils.py:219). */
int gp_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int32_t *idx, gp_stream_t s);
int gp_ball_query_arith(int arith, int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int32_t *idx,
                        gp_stream_t s);

/* group_points_wrapper (group_points.cpp:26-37, group_points_gpu.cu:47-86): points [b,c,n], idx [b,np,ns] -> out [b,c,np,ns]. */
int gp_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int32_t *idx, float *out, gp_stream_t s);
/* group_points_grad_wrapper (group_points_gpu.cu:8-45): grad_out [b,c,np,ns] -> grad_points [b,c,n] (+=, atomic). */
int gp_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int32_t *idx, float *grad_points, gp_stream_t s);

/* three_nn_wrapper (interpolate.cpp, interpolate_gpu.cu:9-74): unknown [b,n,3], known [b,m,3] -> dist2 [b,n,3] f32, idx [b,n,3] i32. */
int gp_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx, gp_stream_t s);
int gp_three_nn_arith(int arith, int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx, gp_stream_t s);
/* three_interpolate_wrapper (interpolate_gpu.cu:77-117): points [b,c,m], idx/weight [b,n,3] -> out [b,c,n]. */
int gp_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx, const float *weight, float *out, gp_stream_t s);
int gp_three_interpolate_arith(int arith, int b, int c, int m, int n, const float *points, const int32_t *idx, const float *weight, float *out,
                               gp_stream_t s);
/* three_interpolate_grad_wrapper (interpolate_gpu.cu:120-160): grad_out [b,c,n] -> grad_points [b,c,m] (+=, atomic). */
int gp_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx, const float *weight, float *grad_points, gp_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * B. Fused PointNet++(MSG) encoder (replaces Pointnet2ClsMSG.forward, pointnet2.py:203-211, and
 *    _PointnetSAModuleBase.forward, pointnet2_modules.py:19-56).  Features are kept POINT-MAJOR
 *    [b, n, C] on the device (the reference keeps [b, C, n]).
 * ------------------------------------------------------------------------------------------------ */

/* FPS + gather for up to 3 consecutive set-abstraction levels in one launch (one workgroup per cloud).
 * xyz [b,n0,3]; level l selects m[l] points out of the previous level's selection.
 * idx_l [b,m_l] i32 (indices into the previous level's point list), new_xyz_l [b,m_l,3].  Unused levels: m = 0. */
int gp_fps_chain(int b, int n0, int nlevels, const int *m, const float *xyz, int32_t *idx0, float *new_xyz0,
                 int32_t *idx1, float *new_xyz1, int32_t *idx2, float *new_xyz2, gp_stream_t s);
int gp_fps_chain_arith(int arith, int b, int n0, int nlevels, const int *m, const float *xyz, int32_t *idx0, float *new_xyz0,
                       int32_t *idx1, float *new_xyz1, int32_t *idx2, float *new_xyz2, gp_stream_t s);

/* Ball query for the two scales of one MSG level in a single pass; rows with no hit are zero-filled. */
int gp_ball_query_msg(int b, int n, int m, float radius0, int nsample0, float radius1, int nsample1, const float *new_xyz,
                      const float *xyz, int32_t *idx0, int32_t *idx1, gp_stream_t s);
int gp_ball_query_msg_arith(int arith, int b, int n, int m, float radius0, int nsample0, float radius1, int nsample1, const float *new_xyz,
                            const float *xyz, int32_t *idx0, int32_t *idx1, gp_stream_t s);

/* One scale of one set-abstraction level: gather neighbourhood -> 3-layer shared MLP (BN folded, ReLU) on
 * fp32 MFMA -> max over the neighbourhood.  Never materialises the grouped [b,C+3,np,ns] tensor.
 *   xyz [b,n,3]; feats_in [b,n,cin] or NULL (cin = 0); new_xyz [b,np,3] and idx [b,np,ns] (grouping mode)
 *   or new_xyz = idx = NULL with np = 1 (GroupAll mode: ns must equal n, absolute xyz, pointnet2_utils.py:268-291).
 *   wpack*: weights packed by gp_pack_weight_size/gp_pack_weight (layer input order = [feats..., dx,dy,dz]);
 *   bias*: folded BN shift per output channel (padded to a multiple of 16).
 *   out [b,np,cout_total]: this scale writes channels [cout_off, cout_off + c3).
 * GroupAll mode accumulates with an integer atomic max (post-ReLU values are >= 0): out must be zeroed first. */
int gp_sa_mlp_max(int b, int n, int np, int ns, int cin, int c1, int c2, int c3, const float *xyz, const float *feats_in,
                  const float *new_xyz, const int32_t *idx, const float *wpack1, const float *bias1, const float *wpack2,
                  const float *bias2, const float *wpack3, const float *bias3, float *out, int cout_total, int cout_off,
                  gp_stream_t s);

/* Hoisted first layer of a grouping level (exact algebra: W1.[feat_j ; xyz_j - c] = W1f.feat_j + W1x.(xyz_j - c)):
 * gp_point_linear computes z[row, 0:n_out] = x[row, 0:k_in] . W1f^T once per SOURCE point (rows = b*n; wpack from gp_pack_weight);
 * gp_sa_pre_mlp_max then gathers z rows (channels [zoff, zoff+c1) of a zstride-wide row; z = NULL for a level without input
 * features), adds wxyz[c][0..2].(xyz_j - centre) + bias1[c], ReLU, and runs layers 2, 3 + max-pool as gp_sa_mlp_max does.
 * wxyz is [round16(c1)][4] (x, y, z, 0), BN scale folded in.  new_xyz = idx = NULL with np = 1, ns = n selects GroupAll
 * (every point once, absolute xyz; out must be zeroed first - the tiles of a cloud combine by integer atomic max).
 * This is the encoder's production path for every level; gp_sa_mlp_max remains as the unhoisted form. */
int gp_point_linear(int rows, int k_in, int n_out, const float *x, const float *wpack, float *z, gp_stream_t s);
int gp_sa_pre_mlp_max(int b, int n, int np, int ns, int c1, int c2, int c3, const float *xyz, const float *new_xyz, const int32_t *idx,
                      const float *z, int zstride, int zoff, const float *wxyz, const float *bias1, const float *wpack2, const float *bias2,
                      const float *wpack3, const float *bias3, float *out, int cout_total, int cout_off, gp_stream_t s);
/* The same with a choice of HIDDEN-LAYER LAYOUT for widths c2 that are not a multiple of 16 (light encoder level 2: 196 = 12 x 16 + 4).
 * GP_SA_TAIL_PLAIN: channel c of the hidden layer is row c of wpack2 / bias2 and column c of wpack3 (what gp_sa_pre_mlp_max assumes).
 * GP_SA_TAIL_SPREAD: the r = c2 % 16 channels of the last, partly filled 16-channel block are moved to positions
 * 4 (c % 4) + c / 4 of that block (gp_sa_tail_position(c2, channel) gives the padded index; the caller packs a [round16(c2), c1]
 * layer-2 matrix / bias and a [c3, round16(c2)] layer-3 matrix with the channels there and zeros elsewhere).  The network is the same
 * function; the register-chain kernels then skip the MFMAs of the last block that only multiply padding (196: one k-step of four). */
#define GP_SA_TAIL_PLAIN 0
#define GP_SA_TAIL_SPREAD 1
int gp_sa_pre_mlp_max_layout(int hidden_layout, int b, int n, int np, int ns, int c1, int c2, int c3, const float *xyz, const float *new_xyz,
                             const int32_t *idx, const float *z, int zstride, int zoff, const float *wxyz, const float *bias1, const float *wpack2,
                             const float *bias2, const float *wpack3, const float *bias3, float *out, int cout_total, int cout_off, gp_stream_t s);
int gp_sa_tail_position(int c2, int channel);

/* OPT-IN, EXPLORATORY (round 5; csrc/sa_bf16x3.hip): gp_sa_pre_mlp_max for the level-2 shapes of the light encoder (c1, c2, c3 = 128, 196, 256;
 * ns = 16 | 32; hoisted first layer: z required) with layers 2 and 3 on the BF16 matrix pipe as three-term split products
 * (a.b ~= a_hi.b_hi + a_lo.b_hi + a_hi.b_lo, fp32 accumulate; relative error ~2^-17 per product instead of 2^-24).  w2_split / w3_split: the
 * weights as hi / lo bf16 pairs in the fragment order of v_mfma_f32_16x16x32_bf16 (genpose_amd/weights.py: pack_bf16x3 -
 * [c1/32][13][2][64][8] and [2][7][8][2][64][8] bf16); bias2 [224] zero padded, channel order as trained.  Never the default: the fp32 entry
 * points above are what every parity claim and the headline bench line run.  b * np * ns must be a multiple of 32. */
int gp_sa_pre_mlp_max_bf16x3(int b, int n, int np, int ns, int c1, int c2, int c3, const float *xyz, const float *new_xyz, const int32_t *idx,
                             const float *z, int zstride, int zoff, const float *wxyz, const float *bias1, const void *w2_split, const float *bias2,
                             const void *w3_split, const float *bias3, float *out, int cout_total, int cout_off, gp_stream_t s);


/* Weight packing for the MFMA layers (host-callable helpers operating on HOST memory):
 * W is [n_out, k_in] row-major (torch Linear / 1x1 conv layout).  Packed size in floats = gp_pack_weight_size(). */
int64_t gp_pack_weight_size(int n_out, int k_in);
int gp_pack_weight(int n_out, int k_in, const float *W, int ldw, float *packed);

/* ------------------------------------------------------------------------------------------------
 * C. Score / energy network and the samplers (scorenet.py:178-222, energynet.py:143-198,
 *    samplers.py:102-160 PC, samplers.py:163-227 PF-ODE with scipy RK45 semantics).
 * ------------------------------------------------------------------------------------------------ */

/* Parameter block shared by the score/energy entry points (all device pointers). */
typedef struct gp_scorenet {
    const float *w_pose0; /* packed [256 x 9]   pose_encoder.0 */
    const float *b_pose0; /* [256] */
    const float *w_pose2; /* packed [256 x 256] pose_encoder.2 */
    const float *b_pose2; /* [256] */
    const float *w_headx; /* packed [768 x 256]: pose_feat columns of fusion_tail_{rot_x,rot_y,trans}.0 stacked */
    const float *w_out;   /* [9 x 256]: rows 0-2 rot_x.2, 3-5 rot_y.2, 6-8 trans.2 */
    const float *b_out;   /* [9] */
    const float *fourier_w; /* [64] t_encoder.0.W */
    const float *w_t1;    /* [128 in][128 out]: t_encoder.1.weight TRANSPOSED */
    const float *b_t1;    /* [128] */
    const float *w_headt; /* [128 in][768 out]: t_feat columns of the three head first layers, TRANSPOSED */
    const float *w_headp; /* packed [768 x 1024]: pts_feat columns of the three head first layers */
    const float *b_head;  /* [768] */
    /* transposed packs for the backward pass of gp_score_div (d score / d pose); unused by every other entry point */
    const float *w_headx_t; /* packed [256 x 768] = w_headx^T */
    const float *w_pose2_t; /* packed [256 x 256] = pose_encoder.2.weight^T */
    const float *w_pose0_t; /* packed [9 x 256]   = pose_encoder.0.weight^T */
} gp_scorenet;

/* cvec[b,768] = W_headp . pts_feat[b] + b_head  (hoisted once per cloud; exact algebra). */
int gp_cloud_embed(int b, const gp_scorenet *net, const float *pts_feat, float *cvec, gp_stream_t s);
/* tvec[nt,768] = W_headt . relu(W_t1 . fourier(t) + b_t1) for nt time values read from DEVICE memory (f32). */
int gp_time_embed(int nt, const gp_scorenet *net, const float *t, float *tvec, gp_stream_t s);
/* ngroups sets of nt time values, set g at t + g * t_stride_floats -> tvec[(g*nt + i), 768] (the grouped RK45 driver reads its stage
 * times straight out of the per-group solver states). */
int gp_time_embed_strided(int nt, int ngroups, int64_t t_stride_floats, const gp_scorenet *net, const float *t, float *tvec, gp_stream_t s);

/* f_theta / score / energy for R = nclouds*k rows.  x [R,9] f32; tvec [768]; sigma = *sigma_dev (f32).
 * mode 0: out[R,9] = f_theta/(sigma+1e-7) (score, scorenet.py:217); mode 1: out[R,2] = IP energy (energynet.py:180-185). */
int gp_score_eval(int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x,
                  const float *sigma_dev, int mode, float *out, gp_stream_t s);

/* Score and the Skilling-Hutchinson divergence estimate of cond_ode_likelihood (samplers.py:49-71) in one launch:
 *   score[R,9] = f_theta(x)/(sigma+1e-7);  div[R] = eps^T (d score / d x) eps  - the reference gets it from
 *   torch.autograd.grad(sum(score * eps), x); here the vector-Jacobian product runs through the transposed weight packs
 *   (ReLU masks from the forward activations kept in LDS).  x, eps [R,9] f32; tvec [768]; sigma = *sigma_dev. */
int gp_score_div(int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x, const float *eps,
                 const float *sigma_dev, float *score, float *div, gp_stream_t s);

/* Score of the ENERGY model (PoseEnergyNet.forward(return_item='score'), energynet.py:200-222): the gradient of the un-decoupled
 * inner-product energy <x, f_theta(x)/sigma> with respect to the pose, which the reference obtains by autograd:
 *   score[R,9] = f_theta/sigma + J_f^T (x/sigma);  energy[R] (may be NULL) = <x, f_theta/sigma>.  `net` = the energy net's block. */
int gp_energy_score(int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x, const float *sigma_dev,
                    float *score, float *energy, gp_stream_t s);

/* gp_score_eval with the launch plan chosen by the caller: tile = 0 (automatic, = gp_score_eval), 16 / 32 / 64 (tile form: one 16-, 32-
 * or 64-row tile per workgroup, activations through LDS), 128 (chain form: 4 waves x 32 rows per workgroup, activations
 * register-resident, weights through an LDS ring - csrc/trunk_chain.h; pays from ~32 000 rows; needs k >= 43 so that the rows of a
 * workgroup span at most 4 clouds, else GP_EINVAL). */
int gp_score_eval_plan(int tile, int nclouds, int k, const gp_scorenet *net, const float *cvec, const float *tvec, const float *x,
                       const float *sigma_dev, int mode, float *out, gp_stream_t s);

/* Rows per workgroup tile of the TILE-form score kernels (the RK45 driver and the backward kernels use the tile form only). */
int gp_score_tile_rows(int nrows);

/* One launch of the predictor-corrector sampler (cond_pc_sampler, samplers.py:102-160), score evaluation fused in.
 * Launch `step` = 0 .. nsteps (nsteps+1 launches, stream order is the only synchronisation):
 *   step > 0      : finishes step-1 for every row - Langevin corrector with the BATCH-MEAN gradient norm
 *                   (samplers.py:130-132; reduced in fixed order from partials[step-1][*]), renormalisation (:142-143),
 *                   Euler-Maruyama predictor with the pre-corrector score (:146-149), normalize_rotation (:152);
 *   step < nsteps : evaluates score(x, t_step) -> score[R,9] and this tile's sum of row norms -> partials[step][tile];
 *   step == nsteps: additionally writes mean_x (+centre, normalised: :157-158).
 * sched [nsteps][4] f32 = {sigma(t_i), g(t_i), step_size, sqrt(step_size)} (host schedule table);
 * tvec_all [nsteps][768] from gp_time_embed; z_* [nsteps][R][9] standard-normal draws; centre [nclouds][3];
 * traj: NULL or [nsteps][R][9] (in-process samples, centre added).
 * partials: [nsteps][ceil(R / gp_score_tile_rows(R))] floats.  gp_pc_step, gp_pc_step_grouped and gp_pc_step_coupled always run the TILE
 * form (one partial per 16- / 32-row workgroup), so this size holds for every R; the chain form of large launches (one partial per
 * wave) is reached through gp_pc_layout + gp_pc_step_plan, whose nparts_out is then the size to allocate. */
int gp_pc_step(int nclouds, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec, const float *tvec_all,
               const float *sched, const float *z_langevin, const float *z_predictor, const float *centre, float *x, float *mean_x,
               float *score, float *partials, float *traj, gp_stream_t s);

/* The same launch over `ngroups` independent batches of `nclouds_per_group` clouds laid out back to back (rows, clouds,
 * noise: group-major).  Everything is row-local except the batch-mean gradient norm, which stays PER GROUP, so every
 * group's result is what gp_pc_step returns for it alone (up to the summation order of the norm partials); serving
 * several batches per launch lets the kernel use 32-row tiles or the chain form (MFMA-bound) where one batch only fills 16-row
 * tiles (weight-stream-bound).  Rows of one group must be a multiple of the tile; partials: [nsteps][ngroups * ceil(rows per group /
 * tile)] with tile = gp_pc_tile_rows() = the TILE-form choice (16 or 32, or GP_EINVAL), which gp_pc_step_grouped / _coupled run and the
 * RK45 driver uses.  (gp_pc_layout + gp_pc_step_plan additionally offer the chain form, with their own partials size.) */
int gp_pc_tile_rows(int ngroups, int nclouds_per_group, int k);
int gp_pc_step_grouped(int ngroups, int nclouds_per_group, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec,
                       const float *tvec_all, const float *sched, const float *z_langevin, const float *z_predictor,
                       const float *centre, float *x, float *mean_x, float *score, float *partials, float *traj, gp_stream_t s);

/* OPT-IN, EXPLORATORY (round 5; csrc/trunk_bf16x3.hip): one launch of the PC sampler - gp_pc_step_grouped's contract (step = 0 .. nsteps, per-group
 * batch-mean coupling, the same buffers) - with the score network's three dense layers on the BF16 matrix pipe as three-term split products
 * (a.b ~= a_hi.b_hi + a_lo.b_hi + a_hi.b_lo, fp32 accumulate).  128 rows per workgroup (k >= 43: a workgroup's rows span at most four clouds;
 * rows per group a multiple of 128 when ngroups > 1), one partial sum per wave: partials [nsteps][*nparts_out of gp_pc_layout_bf16x3].
 * w_*_split: hi / lo bf16 pairs in the fragment order of v_mfma_f32_16x16x32_bf16 (genpose_amd/weights.py: pack_bf16x3 - pose_encoder.0
 * [1][16][2][64][8] in natural k order, pose_encoder.2 [8][16]..., stacked heads [8][48]... in the register chain's k order); b_*, w_out [9][256],
 * b_out [9] fp32.  PC sampler of the score model only (the RK45 driver and every default path keep the fp32 trunk). */
int gp_pc_layout_bf16x3(int ngroups, int nclouds_per_group, int k, int *nparts_out);
int gp_pc_step_bf16x3(int ngroups, int nclouds_per_group, int k, int step, int nsteps, const float *cvec, const float *tvec_all, const float *sched,
                      const float *z_langevin, const float *z_predictor, const float *centre, float *x, float *mean_x, float *score, float *partials,
                      float *traj, const void *w_pose0_split, const void *w_pose2_split, const void *w_headx_split, const float *b_pose0, const float *b_pose2,
                      const float *w_out, const float *b_out, gp_stream_t s);

/* Launch plan of the PC sampler for (ngroups x nclouds_per_group clouds x k candidates) and a model (see gp_pc_step_plan): tile = 0 asks for the automatic choice, else
 * 16 / 32 / 64 / 128 as in gp_score_eval_plan.  *tile_out = the plan taken, *nparts_out = partial sums of |score| per step
 * (`partials` must hold nsteps * nparts floats: one per workgroup in the tile form, one per wave in the chain form).  GP_EINVAL when a
 * workgroup of the plan would straddle two groups.  In the latency regime (score model, tiles x 3 <= CUs) the automatic choice is
 * 16 | GP_PLAN_HEADSPLIT (defined with the RK45 driver below): three workgroups per 16-row tile, one head of the network each; in its
 * `partials` every step keeps, per row, the three heads' sums of squares AND its own copies of the score and the state (nparts = 21 x rows:
 * three workgroups read a tile's state and score and each writes a part, so nothing a launch reads is written by the same launch; `x`
 * keeps the INITIAL state), and it refuses gn_ext (a sharded batch's caller sums per-tile partials: force tile = 16 there). */
int gp_pc_layout(int model, int tile, int ngroups, int nclouds_per_group, int k, int *tile_out, int *nparts_out);
/* gp_pc_step_coupled with the plan chosen by the caller (tile as above; 0 = automatic; every launch of one chain must use the same
 * plan) and the MODEL whose score drives the sampler: 0 = the score network (f / (sigma + 1e-7), scorenet.py:217); 1 = the ENERGY
 * network (`net` = its parameter block): the reference samples from it with the autograd gradient of its inner-product energy
 * (posenet.py:94-130 with PoseEnergyNet.forward(return_item='score'), energynet.py:200-222) - here the forward pass and the
 * vector-Jacobian product run inside the step kernel (tile = 16: through LDS, csrc/score_bwd.h; tile = 128: the register-resident chain
 * form, csrc/trunk_chain_vjp.h, which large launches take), so the energy model gets the same one-graph launch chain.
 * gn_ext / gn_rows_total: as gp_pc_step_coupled when gn_rows_total = 0 (gn_ext = the mean); with gn_rows_total > 0, gn_ext [nsteps][ngroups]
 * holds the SUM of |score| over all rows of the batch (this rank's per-group sum of `partials`, all-reduced in place) and
 * gn_rows_total that row count: one device-side sum and one all-reduce per step, both capturable in the sampler's hipGraph. */
int gp_pc_step_plan(int model, int tile, int ngroups, int nclouds_per_group, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec,
                    const float *tvec_all, const float *sched, const float *z_langevin, const float *z_predictor, const float *centre, float *x,
                    float *mean_x, float *score, float *partials, float *traj, const float *gn_ext, int gn_rows_total, gp_stream_t s);

/* The same launch with the batch-mean gradient norm SUPPLIED: gn_ext [nsteps][ngroups] (device) holds, for step i, the mean of
 * |score_i| over ALL rows of the batch each group belongs to.  For a batch that is sharded over several GPUs (SURVEY §8e caveat): the
 * host sums this rank's `partials` of step i, all-reduces the sum across the ranks and writes gn_ext[i] before launching step i+1, so
 * every shard takes the Langevin step size the unsharded batch would take (samplers.py:130-132).  gn_ext == NULL: gp_pc_step_grouped. */
int gp_pc_step_coupled(int ngroups, int nclouds_per_group, int k, int step, int nsteps, const gp_scorenet *net, const float *cvec,
                       const float *tvec_all, const float *sched, const float *z_langevin, const float *z_predictor,
                       const float *centre, float *x, float *mean_x, float *score, float *partials, float *traj, const float *gn_ext,
                       gp_stream_t s);

/* Probability-flow ODE sampler = cond_ode_sampler (samplers.py:163-227) over scipy's RK45 (rk.py / common.py):
 * Dormand-Prince 5(4), f64 state and controller resident in device memory, f32 score network, batch-global RMS
 * error norm, SAFETY 0.9 / MIN_FACTOR 0.2 / MAX_FACTOR 10, Hairer initial step.
 *   state   : opaque device block of gp_rk45_state_bytes() bytes (field offsets: gp_rk45_state_layout)
 *   y, ynew : [R*9] f64;  K : [7][R*9] f64;  partials : [3][ceil(R/tile)] f64;  tvec : [8][768] f32
 *   traj    : NULL or [traj_cap][R*9] f64 (accepted states; slot 0 = y0)
 * gp_rk45_phase launches one fixed, capture-safe sequence per call:
 *   0 reset(t0 -> t_bound, rtol, atol)   1 f0 + d0,d1 -> h0   2 f1 + d2 -> h_abs, first stage times
 *   3 ONE attempt: 6 fused stage kernels (stage update + score evaluation) + controller (+ trajectory record);
 *     a no-op once the device-side status word is non-zero
 *   4 set evaluation slot 0 to time `t0` (the denoise evaluation at eps)
 *   5 finish: denoise step (samplers.py:209-218) with scale `denoise_scale`, normalize_rotation, + centre -> x_out [R,9] f64,
 *     and the same post-processing of the first `nstates` trajectory states.
 * tvec is scratch owned by the solver: every kernel that decides stage times (reset, step controller, phase 4) writes their time
 * embeddings there itself (the arithmetic of gp_time_embed), so no launch separates the controller from the next stage kernel. */
int64_t gp_rk45_state_bytes(void);
/* Dense-output mode = solve_ivp(..., t_eval=np.linspace(T0, eps, n)) (samplers.py:201-205): call after phase 0 with traj = NULL
 * there; t_eval_dev [n_eval] f64 on the device, P_host = RK45's 7x4 dense-output matrix (row-major, HOST memory).  traj
 * [n_eval][R*9] then receives the 4th-order interpolant at every t_eval point (scipy RkDenseOutput). */
int gp_rk45_set_dense(void *state, const double *t_eval_dev, int n_eval, const double *P_host, gp_stream_t s);
int gp_rk45_state_layout(int64_t *offsets, int n); /* n >= 13: t,h_abs,status,n_attempts,n_accepted,nfev,err_norm,log_t,log_h,log_err,log_acc,stage_t,last_accepted */
int gp_rk45_phase(int phase, int nclouds, int k, const gp_scorenet *net, const float *cvec, float *tvec, const float *centre,
                  void *state, double *y, double *ynew, double *K, double *partials, double *traj, int traj_cap, double t0,
                  double t_bound, double rtol, double atol, double denoise_scale, int do_denoise, int nstates, double *x_out,
                  gp_stream_t s);

/* The same driver over `ngroups` independent batches that share every launch while keeping their OWN step controllers (error norm,
 * accept / reject and step size per group, exactly as separate solve_ivp calls); a finished group's workgroups exit at once.
 * Rows, clouds and solver states are laid out group-major: state = ngroups * gp_rk45_state_bytes(), tvec [ngroups][8][768]
 * (scratch), partials [3][nblocks] with
 * nblocks = ngroups * ceil(rows_per_group / tile), tile = gp_pc_tile_rows(ngroups, nclouds_per_group, k) (rows of a group must be
 * a multiple of it).  traj: every group writes its own rows at its own accepted-step slot. */
int gp_rk45_phase_grouped(int phase, int ngroups, int nclouds_per_group, int k, const gp_scorenet *net, const float *cvec, float *tvec,
                          const float *centre, void *state, double *y, double *ynew, double *K, double *partials, double *traj, int traj_cap,
                          double t0, double t_bound, double rtol, double atol, double denoise_scale, int do_denoise, int nstates, double *x_out,
                          gp_stream_t s);
/* The same driver for the three right-hand sides it can integrate (`model`):
 *   0  gp_rk45_phase_grouped: probability-flow ODE of the score network;
 *   1  the same ODE driven by the ENERGY network's score, the gradient of its inner-product energy (posenet.py:94-130 on a
 *      PoseEnergyNet, energynet.py:200-222; `net` = the energy net's block) - forward + vector-Jacobian product inside the stage kernel;
 *   2  the likelihood ODE of cond_ode_likelihood (samplers.py:22-99): state [R][10] = pose and accumulated log-density change,
 *      d logp / dt = -g^2/2 probe^T (d score / d x) probe with the fixed Skilling-Hutchinson `probe` [R][9] (device); y, ynew [R*10],
 *      K [7][R*10]; one error norm over all R*10 components (scipy integrates the concatenated vector); phase 5 copies the final
 *      state to x_out [R][10] (no denoise / normalisation); phases 4 and the trajectory arguments are unused.
 * Models 1 and 2 run on 16-row tiles: partials [3][ngroups * ceil(rows_per_group / 16)].
 * A batch SHARDED over several GPUs (SURVEY §8e caveat: scipy's error norm runs over the whole batch): ext_sums [2][ngroups] (device,
 * f64) and ext_rows_per_group = rows of a group over all ranks.  Phases 1, 2, 3 then stop after writing this rank's per-group sums of
 * squares to ext_sums; the caller all-reduces ext_sums (RCCL: capturable with the launches) and runs phase 11, 12 or 13 = the step
 * controller on the reduced sums (+ trajectory record).  Every shard then takes the accept / reject sequence of the unsharded batch.
 * ext_sums = NULL: the controller reduces the local partials itself (phases 11-13 are GP_EINVAL).
 * plan: rows per workgroup of the stage kernels - 16 / 32 / 64 = tile form (models 1 and 2, which need the backward pass: 16 only), 128 = the
 * chain form of the trunk (every model; k >= 43, rows_per_group % 128 == 0 when ngroups > 1); 16 | GP_PLAN_HEADSPLIT = the latency
 * regime's head-split plan (score model): THREE workgroups per 16-row tile, each recomputing pose_encoder and evaluating one head
 * (scorenet.py:178-222: the three fusion tails are independent given the pose features) - half the weight stream and half the MFMA issue
 * per workgroup, 3x the CUs; every stage of an attempt is then a launch of its own (a stage reads all nine components of the previous
 * one); recommended by gp_rk45_plan_rows() while tiles x 3 <= CUs (gp_plan_headsplit_pays).  0 = a whole-tile plan is picked.
 * partials: gp_rk45_partials_count() doubles ([3][ngroups * ceil(rows_per_group / rows per workgroup) * (3 under the head-split plan)]). */
#define GP_PLAN_HEADSPLIT 0x100
/* 16 | GP_PLAN_SHARED or 48 | GP_PLAN_SHARED (round 6; score model, one group; picked by gp_rk45_plan_rows() when it applies): the SHARED-CHUNK plan for
 * launches whose 16-row chunks do not divide over the CUs - T chunks on C CUs with T / C = 1 or 3 and 6 (T mod C) <= C, e.g. the 12 800 rows
 * of scripts/eval_single.sh's batches (800 chunks on 256 CUs).  An attempt is ONE launch of C workgroups: each owns T / C whole chunks for
 * all six stages, and the 6 x (T mod C) (chunk, stage) units of the left-over chunks are dealt out one per workgroup, so the busiest CU
 * evaluates 6 x (T / C) + 1 chunk-stages per attempt instead of 6 x (T / C + 1) (csrc/rk45.hip: rk45_attempt_shared_kernel).  The other
 * phases run on the whole-tile plan.  Results: every row's right-hand side is the 32- / 64-row tile plans' bit for bit (the four-wave tiles form
 * the output sums in their order, score_trunk.h ORDER8); the error norm's partial sums add in another order (poses within 1e-13 of theirs). */
#define GP_PLAN_SHARED 0x200
int gp_plan_headsplit_pays(int ntiles16); /* 1 while three workgroups per 16-row tile still get a CU each */
/* The plan the driver recommends for a launch (may carry GP_PLAN_HEADSPLIT / GP_PLAN_SHARED) and the number of doubles `partials` must hold
 * under a plan (plan = 0: what gp_rk45_phase_model resolves 0 to - always a WHOLE-tile plan, so that a buffer of 3 x ceil(rows / 16) doubles
 * per group, the size the entry points without a plan argument document, stays sufficient; the head-split and shared-chunk plans are taken
 * only when passed explicitly - ABI note in INTEGRATION.md section 3). */
int gp_rk45_plan_rows(int model, int ngroups, int nclouds_per_group, int k);
int gp_rk45_plan_rows_unshared(int model, int ngroups, int nclouds_per_group, int k); /* the same without GP_PLAN_SHARED (ext_sums callers) */
int gp_rk45_partials_count(int model, int plan, int ngroups, int nclouds_per_group, int k);
int gp_rk45_phase_model(int model, int plan, const float *probe, int phase, int ngroups, int nclouds_per_group, int k, const gp_scorenet *net, const float *cvec,
                        float *tvec, const float *centre, void *state, double *y, double *ynew, double *K, double *partials, double *traj,
                        int traj_cap, double t0, double t_bound, double rtol, double atol, double denoise_scale, int do_denoise, int nstates,
                        double *x_out, double *ext_sums, int ext_rows_per_group, gp_stream_t s);
/* Ragged variant: groups with different numbers of clouds (tracking: the objects of one frame form a group, frames of different
 * sequences share the launches).  grp_info [ngroups][4] = {first workgroup, workgroups, rows, first row}; blk_info [nblocks][3] =
 * {group, first row, end row (exclusive) of the group} per workgroup of `tile` (16 or 32) rows; both device int32.  Rows stay
 * cloud-major (k rows per cloud), groups occupy consecutive row ranges. */
int gp_rk45_phase_ragged(int phase, int ngroups, const int32_t *grp_info, int nblocks, const int32_t *blk_info, int tile, int nclouds_total, int k,
                         const gp_scorenet *net, const float *cvec, float *tvec, const float *centre, void *state, double *y, double *ynew,
                         double *K, double *partials, double *traj, int traj_cap, double t0, double t_bound, double rtol, double atol,
                         double denoise_scale, int do_denoise, int nstates, double *x_out, gp_stream_t s);
int gp_rk45_set_dense_grouped(int ngroups, void *state, const double *t_eval_dev, int n_eval, const double *P_host, gp_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * D. Ranking and aggregation (reward.py:131-155, sgpa_utils.py:897-954, evaluation_tracking.py:60-77)
 * ------------------------------------------------------------------------------------------------ */

/* poses [b,k,9] f32/f64 (is_f64), energy [b,k,2] f32 -> sorted_poses (same dtype), sorted_energy [b,k,2],
 * order [b,k,2] i32 (stable descending), avg_pose [b,7] f32 (w,x,y,z,tx,ty,tz) over the top `sel` candidates. */
int gp_rank_aggregate(int b, int k, int sel, int is_f64, const void *poses, const float *energy, void *sorted_poses,
                      float *sorted_energy, int32_t *order, float *avg_pose, gp_stream_t s);
/* The same launch with the 4x4 forms the runners hand on written by it as well (either may be NULL): sorted_rt [b,k,4,4] f64 =
 * gp_pose9_to_rt(sorted_poses), avg_rt [b,4,4] f32 = gp_quat_trans_to_rt(avg_pose) - bit for bit; the ranking step of a tracking frame
 * (evaluation_tracking.py:316-330: energies -> sort -> average) is then one launch behind the energy evaluation. */
int gp_rank_aggregate_rt(int b, int k, int sel, int is_f64, const void *poses, const float *energy, void *sorted_poses,
                         float *sorted_energy, int32_t *order, float *avg_pose, double *sorted_rt, float *avg_rt, gp_stream_t s);

/* The 4x4 homogeneous matrices the runners hand on (evaluation_single.py:325-332, evaluation_tracking.py:60-77), one launch each instead of
 * ~15 tensor operations: poses [n][9] (f32 or f64: is_f64) -> out [n][4][4] f64 (Gram-Schmidt of the two rotation columns in f64, as
 * get_rot_matrix on float64 rows); quat_trans [n][7] f32 (w, x, y, z, t - gp_rank_aggregate's avg_pose) -> out [n][4][4] f32 (pytorch3d
 * quaternion_to_matrix). */
int gp_pose9_to_rt(int n, int is_f64, const void *pose, double *out, gp_stream_t s);
int gp_quat_trans_to_rt(int n, const float *quat_trans, float *out, gp_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* GENPOSE_HIP_H */
