"""Depth + mask -> point cloud pre-processing.
CPU: the oracle (oracle/preprocess_oracle.py, oracle/cv2_restated.py) against fixture G11, which was produced by composing the
reference's own get_bbox / get_2d_coord_np / crop_resize_by_warp_affine (oracle/gen_golden.py --g11); the product's host
arithmetic (window, inverse crop map) against the oracle.  GPU: the HIP path against the oracle, bit for bit."""
import hashlib

import numpy as np
import pytest

from genpose_amd import preprocess as pp
from genpose_amd import synth
from oracle import cv2_restated as cv2r
from oracle import preprocess_oracle as po


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def test_oracle_matches_reference_fixture(golden):
    g = golden("g11_preprocess.npz")
    depth, masks, rois, class_ids = synth.golden_depth_frame()
    K = g["intrinsics"]
    assert np.array_equal(K, po.REAL_INTRINSICS)
    for i in range(len(class_ids)):
        assert tuple(g[f"i{i}_bbox"]) == po.get_bbox(rois[i])
        cloud = po.instance_cloud(depth, masks[:, :, i], rois[i], K)
        if f"i{i}_cloud_sha" not in g:
            assert cloud is None
            continue
        assert cloud.shape[0] == int(g[f"i{i}_n_valid"]) and cloud.dtype == np.float32
        np.testing.assert_array_equal(cloud[:64], g[f"i{i}_cloud_head"])
        np.testing.assert_array_equal(_sha(cloud), g[f"i{i}_cloud_sha"])
    pts, cat, inst = po.frame_clouds(depth, masks, rois, class_ids, K, rng=np.random.RandomState(11))
    np.testing.assert_array_equal(pts, g["points"])
    assert cat == list(g["cat_id"]) and inst == list(g["valid_inst"])
    assert sorted(set(range(len(class_ids))) - set(inst)) == [4, 5]  # the empty mask and the depth hole are skipped


def test_host_window_and_inverse_map_equal_oracle():
    rng = np.random.default_rng(0)
    for _ in range(300):
        y1, x1 = int(rng.integers(0, 470)), int(rng.integers(0, 630))
        roi = [y1, x1, min(479, y1 + int(rng.integers(1, 470))), min(639, x1 + int(rng.integers(1, 630)))]
        win = pp.get_bbox(roi)
        assert win == po.get_bbox(roi)
        rmin, rmax, cmin, cmax = win
        center = np.array([0.5 * (cmin + cmax), 0.5 * (rmin + rmax)])
        scale = min(max(rmax - rmin, cmax - cmin), 640) * 1.0
        ref_inv = cv2r.invertAffineTransform(po.get_affine_transform(center, scale, 256))
        mine = pp.inverse_crop_map(win, 256, 480, 640)
        np.testing.assert_allclose(mine, ref_inv, rtol=0, atol=1e-9)
        # and the sampled source pixels are identical
        sx, sy = cv2r.nearest_source_index(po.get_affine_transform(center, scale, 256), (256, 256))
        j = np.arange(256)
        np.testing.assert_array_equal(sx[0], np.floor(j * scale / 256 + mine[0, 2] + 0.5).astype(np.int64))
        np.testing.assert_array_equal(sy[:, 0], np.floor(j * scale / 256 + mine[1, 2] + 0.5).astype(np.int64))


def test_warp_nearest_border_and_dtype():
    img = np.arange(20, dtype=np.uint16).reshape(4, 5)
    M = np.array([[1.0, 0.0, 2.0], [0.0, 1.0, -1.0]])  # dst = src + (2, -1)
    out = cv2r.warpAffine(img, M, (5, 4), flags=cv2r.INTER_NEAREST)
    assert out.dtype == np.uint16 and out.shape == (4, 5)
    exp = np.zeros((4, 5), dtype=np.uint16)
    exp[0:3, 2:5] = img[1:4, 0:3]
    np.testing.assert_array_equal(out, exp)


def test_sample_points_rules():
    pcl = np.arange(15, dtype=np.float32).reshape(5, 3)
    np.testing.assert_array_equal(po.sample_points(pcl, 12)[:, 0] // 3, np.arange(12) % 5)
    assert po.sample_points(pcl, 5) is pcl
    sub = po.sample_points(pcl, 3, np.random.RandomState(0))
    np.testing.assert_array_equal(sub, pcl[np.random.RandomState(0).permutation(5)[:3]])


@pytest.mark.gpu
def test_hip_clouds_bit_exact_vs_oracle():
    import torch
    for seed in (5, 6):
        depth, masks, rois, class_ids = synth.golden_depth_frame(seed)
        d2c = pp.DepthToClouds(pp.REAL_INTRINSICS)
        pcl, count, dcount = d2c.full_clouds(depth, masks, rois)
        torch.cuda.synchronize()
        for i in range(len(class_ids)):
            ref = po.instance_cloud(depth, masks[:, :, i], rois[i], po.REAL_INTRINSICS)
            c = int(count[i])
            if ref is None:
                assert c <= 1 or int(dcount[i]) <= 1
                continue
            assert c == ref.shape[0]
            np.testing.assert_array_equal(pcl[i, :c].cpu().numpy(), ref)  # float32, bit for bit, same point order
        got = d2c(depth, masks, rois, class_ids, rng=np.random.RandomState(11))
        pts, cat, inst = po.frame_clouds(depth, masks, rois, class_ids, po.REAL_INTRINSICS, rng=np.random.RandomState(11))
        np.testing.assert_array_equal(got["points"].cpu().numpy(), pts)
        assert got["cat_id"] == cat and got["valid_inst"] == inst


@pytest.mark.gpu
def test_hip_clouds_feed_the_runner():
    """Pre-processing output goes straight into the agents: clouds stay on the device."""
    import torch
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.weights_synth import make_state_dict
    depth, masks, rois, class_ids = synth.golden_depth_frame()
    got = pp.DepthToClouds(pp.REAL_INTRINSICS)(depth, masks, rois, class_ids, rng=np.random.RandomState(1))
    agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=8))
    agent.load_state_dict(make_state_dict(0, "score"))
    pts = got["points"]
    pred = agent.pred_func({"pts": pts, "pts_center": pts.mean(dim=1)}, repeat_num=4, save_path=None)
    assert pred.shape == (len(got["valid_inst"]), 4, 9) and torch.isfinite(pred).all()
