"""Depth image + Mask-RCNN detections -> per-instance 1024-point clouds on the GPU.

Replaces the per-frame body of detect_mrcnn_genpose (runners/evaluation_single.py:140-216): for every detection the square
crop window of get_bbox (utils/sgpa_utils.py:214-242), the 256 x 256 nearest-neighbour crops of depth / mask / pixel
coordinates (utils/datasets_utils.py:82-136), depth_to_pcl and sample_points.  Input formats are the reference's: `depth`
uint16 millimetres [H,W] (load_depth, sgpa_utils.py:194-211), `masks` bool [H,W,n], `rois` [n,4] (y1,x1,y2,x2), `class_ids` [n].
Output: the `valid_pts` / `cat_id` / `valid_inst` entries of the reference's per-image record (evaluation_single.py:241-253),
with the clouds left on the device for the pose agents.
"""
import numpy as np
import torch

from . import _lib
from ._lib import ptr, stream_ptr

REAL_INTRINSICS = np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]], dtype=np.float32)   # evaluation_single.py:54
CAMERA_INTRINSICS = np.array([[577.5, 0, 319.5], [0, 577.5, 239.5], [0, 0, 1]], dtype=np.float32)            # evaluation_single.py:50


def get_bbox(bbox, img_h=480, img_w=640):
    """Square crop window (rmin, rmax, cmin, cmax): side = the next multiple of 40 above the box (<= 440), centred on the
    box and shifted back inside the image."""
    y1, x1, y2, x2 = [int(v) for v in bbox]
    half = int(min((max(y2 - y1, x2 - x1) // 40 + 1) * 40, 440) / 2)
    cy, cx = (y1 + y2) // 2, (x1 + x2) // 2
    rmin, rmax, cmin, cmax = cy - half, cy + half, cx - half, cx + half
    if rmin < 0:
        rmin, rmax = 0, rmax - rmin
    if cmin < 0:
        cmin, cmax = 0, cmax - cmin
    if rmax > img_h:
        rmin, rmax = rmin - (rmax - img_h), img_h
    if cmax > img_w:
        cmin, cmax = cmin - (cmax - img_w), img_w
    return rmin, rmax, cmin, cmax


def inverse_crop_map(window, img_size, im_h, im_w):
    """Destination -> source map of crop_resize_by_warp_affine for a get_bbox window: get_affine_transform (rot = 0) maps the
    window centre to the crop centre with gain img_size / scale, scale = min(window side, max(H, W)); its inverse is
    source = dest * scale / img_size + (centre - scale / 2).  [2,3] float64."""
    rmin, rmax, cmin, cmax = window
    scale = float(min(max(rmax - rmin, cmax - cmin), max(im_h, im_w)))
    g = scale / float(img_size)
    cx, cy = 0.5 * (cmin + cmax), 0.5 * (rmin + rmax)
    return np.array([[g, 0.0, cx - 0.5 * scale], [0.0, g, cy - 0.5 * scale]], dtype=np.float64)


class DepthToClouds:
    def __init__(self, intrinsics=REAL_INTRINSICS, n_pts=1024, img_size=256, device="cuda"):
        _lib.check_device()
        K = np.asarray(intrinsics, dtype=np.float32).reshape(-1)
        self.fx, self.fy, self.cx, self.cy = float(K[0]), float(K[4]), float(K[2]), float(K[5])
        self.n_pts, self.img, self.dev = n_pts, img_size, torch.device(device)

    def full_clouds(self, depth, masks, rois):
        """All valid points of every detection, raster order of the crop: (pcl [n,img*img,3] f32, count [n], depth_count [n])."""
        depth = torch.as_tensor(np.ascontiguousarray(depth)) if not torch.is_tensor(depth) else depth
        if depth.dtype not in (torch.uint16, torch.int16):
            raise RuntimeError("depth must be uint16 millimetres")
        H, W = depth.shape
        masks = torch.as_tensor(np.ascontiguousarray(masks)) if not torch.is_tensor(masks) else masks
        n = masks.shape[2] if masks.dim() == 3 else 0
        if n != len(rois):
            raise ValueError(f"{n} masks for {len(rois)} rois")
        depth = depth.to(self.dev).contiguous()
        masks = masks.to(self.dev).to(torch.uint8).contiguous()
        minv = np.stack([inverse_crop_map(get_bbox(r, H, W), self.img, H, W) for r in rois]).reshape(n, 6) if n else np.zeros((0, 6))
        minv = torch.from_numpy(minv).to(self.dev)
        cap = self.img * self.img
        pcl = torch.empty(n, cap, 3, device=self.dev)
        count = torch.zeros(n, dtype=torch.int32, device=self.dev)
        dcount = torch.zeros(n, dtype=torch.int32, device=self.dev)
        _lib.call("gp_roi_to_cloud", H, W, n, self.img, ptr(depth), ptr(masks), ptr(minv), self.fx, self.fy, self.cx, self.cy, ptr(pcl),
                  ptr(count), ptr(dcount), stream_ptr())
        return pcl, count, dcount

    def __call__(self, depth, masks, rois, class_ids, rng=np.random):
        """-> dict(points [k,n_pts,3] device f32, cat_id [k] list, valid_inst [k] list) for the k instances the reference keeps."""
        pcl, count, dcount = self.full_clouds(depth, masks, rois)
        c, d = count.cpu().numpy(), dcount.cpu().numpy()
        keep = [i for i in range(len(c)) if d[i] > 1 and c[i] > 1]  # evaluation_single.py:201-208
        if not keep:
            return {"points": torch.zeros(0, self.n_pts, 3, device=self.dev), "cat_id": [], "valid_inst": []}
        ids = np.zeros((len(keep), self.n_pts), dtype=np.int32)
        for k, i in enumerate(keep):  # same draw order as the reference's per-instance np.random.permutation
            if c[i] > self.n_pts:
                ids[k] = rng.permutation(int(c[i]))[: self.n_pts]
        sel = torch.as_tensor(keep, device=self.dev)
        pcl_k, cnt_k = pcl[sel].contiguous(), count[sel].contiguous()
        ids_t = torch.from_numpy(ids).to(self.dev)
        out = torch.empty(len(keep), self.n_pts, 3, device=self.dev)
        _lib.call("gp_cloud_sample", len(keep), pcl.shape[1], self.n_pts, ptr(pcl_k), ptr(cnt_k), ptr(ids_t), ptr(out), stream_ptr())
        return {"points": out, "cat_id": [int(class_ids[i]) - 1 for i in keep], "valid_inst": keep}
