"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's depth + mask -> point cloud pre-processing
(SURVEY §8f row 1): the per-instance body of detect_mrcnn_genpose (runners/evaluation_single.py:105-216) with
get_bbox (utils/sgpa_utils.py:214-242), get_2d_coord_np / crop_resize_by_warp_affine / get_affine_transform
(utils/datasets_utils.py:4-22, 82-136).  The product (genpose_amd/preprocess.py + csrc/preprocess.hip) never imports this.

Pinned against the reference's own functions where they are importable (get_bbox, get_2d_coord_np, crop_resize_by_warp_affine,
get_affine_transform - fixture G11, oracle/gen_golden.py --g11); depth_to_pcl and sample_points are nested functions of
detect_mrcnn_genpose (not importable) and are restated from the source lines cited; the two cv2 calls underneath are
restated in oracle/cv2_restated.py: parity unpinned at the cv2 level.
"""
import numpy as np

from . import cv2_restated as cv2r

REAL_INTRINSICS = np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]], dtype=np.float32)  # evaluation_single.py:54
CAMERA_INTRINSICS = np.array([[577.5, 0, 319.5], [0, 577.5, 239.5], [0, 0, 1]], dtype=np.float32)           # evaluation_single.py:50


def get_bbox(bbox):
    """sgpa_utils.py:214-242: square crop window, side = multiple of 40 (<= 440), pushed back inside the 480 x 640 image."""
    y1, x1, y2, x2 = [int(v) for v in bbox]
    img_width, img_length = 480, 640
    window = min((max(y2 - y1, x2 - x1) // 40 + 1) * 40, 440)
    cy, cx = (y1 + y2) // 2, (x1 + x2) // 2
    rmin, rmax = cy - int(window / 2), cy + int(window / 2)
    cmin, cmax = cx - int(window / 2), cx + int(window / 2)
    if rmin < 0:
        rmax += -rmin
        rmin = 0
    if cmin < 0:
        cmax += -cmin
        cmin = 0
    if rmax > img_width:
        rmin -= rmax - img_width
        rmax = img_width
    if cmax > img_length:
        cmin -= cmax - img_length
        cmax = img_length
    return rmin, rmax, cmin, cmax


def get_affine_transform(center, scale, output_size):
    """datasets_utils.py:96-136 with rot = 0, shift = 0, inv = False."""
    center = np.asarray(center)
    src_w, dst_w, dst_h = float(scale), float(output_size), float(output_size)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = center + np.array([0.0, src_w * -0.5])
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + np.array([0, dst_w * -0.5], np.float32)
    third = lambda a, b: b + np.array([-(a - b)[1], (a - b)[0]], dtype=np.float32)
    src[2, :] = third(src[0, :], src[1, :])
    dst[2, :] = third(dst[0, :], dst[1, :])
    return cv2r.getAffineTransform(np.float32(src), np.float32(dst))


def crop_resize_nearest(img, center, scale, output_size):
    return cv2r.warpAffine(img, get_affine_transform(center, scale, output_size), (int(output_size), int(output_size)), flags=cv2r.INTER_NEAREST)


def instance_cloud(raw_depth, inst_mask, roi, intrinsics, img_size=256):
    """One detection -> all valid points in raster order of the 256 x 256 crop, metres, float32 [n,3]; None when the
    reference skips the instance (<= 1 valid depth or <= 1 valid masked pixel, evaluation_single.py:201-208)."""
    im_H, im_W = raw_depth.shape
    rmin, rmax, cmin, cmax = get_bbox(roi)
    mask = np.logical_and(inst_mask, raw_depth > 0)
    x = np.linspace(0, im_W - 1, im_W, dtype=np.float32)
    y = np.linspace(0, im_H - 1, im_H, dtype=np.float32)
    coord_2d = np.asarray(np.meshgrid(x, y)).transpose(1, 2, 0)  # get_2d_coord_np(...).transpose(1, 2, 0): [H,W,(x,y)]
    x1, y1, x2, y2 = cmin, rmin, cmax, rmax
    center = np.array([0.5 * (x1 + x2), 0.5 * (y1 + y2)])
    scale = min(max(y2 - y1, x2 - x1), max(im_H, im_W)) * 1.0
    roi_coord = crop_resize_nearest(coord_2d, center, scale, img_size).transpose(2, 0, 1)
    roi_mask = crop_resize_nearest(mask.copy().astype(np.float32), center, scale, img_size)[None]
    roi_depth = crop_resize_nearest(raw_depth, center, scale, img_size)[None]
    depth_valid = roi_depth > 0
    if np.sum(depth_valid) <= 1.0:
        return None
    if np.sum(roi_mask.astype(np.bool_) * depth_valid) <= 1.0:
        return None
    # depth_to_pcl (evaluation_single.py:107-118), all float32
    K = intrinsics.reshape(-1)
    cx, cy, fx, fy = K[2], K[5], K[0], K[4]
    depth = roi_depth.reshape(-1).astype(np.float32)
    valid = ((depth > 0) * roi_mask.reshape(-1)) > 0
    depth = depth[valid]
    x_map = roi_coord[0].reshape(-1)[valid]
    y_map = roi_coord[1].reshape(-1)[valid]
    real_x = (x_map - cx) * depth / fx
    real_y = (y_map - cy) * depth / fy
    pcl = np.stack((real_x, real_y, depth), axis=-1).astype(np.float32)
    return pcl / 1000.0


def sample_points(pcl, n_pts, rng=np.random):
    """evaluation_single.py:120-133: tile when short, a random subset (first n of a permutation) when long."""
    total = pcl.shape[0]
    if total < n_pts:
        return np.concatenate([np.tile(pcl, (n_pts // total, 1)), pcl[: n_pts % total]], axis=0)
    if total > n_pts:
        return pcl[rng.permutation(total)[:n_pts]]
    return pcl


def frame_clouds(raw_depth, masks, rois, class_ids, intrinsics, n_pts=1024, rng=np.random, img_size=256):
    """Frame-level loop (evaluation_single.py:162-216) -> (points [k,n_pts,3], cat_id list, valid_inst list)."""
    pts, cat, inst = [], [], []
    for i in range(len(class_ids)):
        cloud = instance_cloud(raw_depth, masks[:, :, i], rois[i], intrinsics, img_size)
        if cloud is None:
            continue
        pts.append(sample_points(cloud, n_pts, rng))
        cat.append(int(class_ids[i]) - 1)
        inst.append(i)
    return (np.stack(pts) if pts else np.zeros((0, n_pts, 3), dtype=np.float32)), cat, inst
