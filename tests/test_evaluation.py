"""Host-side mAP evaluation (genpose_amd/evaluation.py) against the reference's own compute_mAP and its
building blocks (fixture G10: tests/golden/g10_map.npz, made by oracle/gen_golden.py --g10 from the imported reference)."""
import copy

import numpy as np
import pytest

from genpose_amd import evaluation as ev
from genpose_amd import synth

DEGREE = list(range(0, 46, 1))
SHIFT = [i / 2 for i in range(21)]
IOU = [i / 100 for i in range(101)]


@pytest.fixture(scope="module")
def data(golden):
    g = golden("g10_map.npz")
    return g, synth.golden_map_results(77, 12, int(g["K"]))


def _first_rich(results):
    return next(x for x in results if len(x["gt_class_ids"]) >= 2 and len(x["pred_class_ids"]) >= 2)


def test_building_blocks(data):
    g, results = data
    r = _first_rich(results)
    gm, pm, ov, idx = ev.compute_2d_IoU_matches(r["gt_class_ids"], r["gt_bboxes"], r["pred_class_ids"], r["pred_bboxes"], r["pred_scores"], IOU)
    np.testing.assert_array_equal(gm, g["blk_gt_matches"])
    np.testing.assert_array_equal(pm, g["blk_pred_matches"])
    np.testing.assert_array_equal(ov, g["blk_overlaps"])  # float32 IoUs, bit for bit
    np.testing.assert_array_equal(idx, g["blk_indices"])
    sel, avg, sel_e = ev.sort_sRT_by_energy(r["multi_hypothesis_pred_RTs"].copy(), r["energy"].copy(), None, "energy_ranker", 0.6, "average")
    np.testing.assert_array_equal(sel, g["blk_selected"])
    np.testing.assert_array_equal(sel_e, g["blk_selected_energy"])
    np.testing.assert_allclose(avg, g["blk_average"], rtol=0, atol=2e-6)  # quaternion average runs through a 4x4 eigh
    rt = ev.compute_RT_overlaps(r["gt_class_ids"], r["gt_RTs"], r["gt_handle_visibility"], r["pred_class_ids"], g["blk_average"])
    np.testing.assert_allclose(rt, g["blk_RT_overlaps"], rtol=1e-9, atol=1e-9)
    pgm, ppm = ev.compute_RT_matches(g["blk_RT_overlaps"], r["pred_class_ids"], r["gt_class_ids"], DEGREE + [360], SHIFT + [100])
    np.testing.assert_array_equal(pgm, g["blk_pose_gt_matches"])
    np.testing.assert_array_equal(ppm, g["blk_pose_pred_matches"])


def test_random_ranker(data):
    """ranker='random' (utils/sgpa_utils.py:926-927): the hypotheses are ranked by np.random.rand(pred_num, repeat_num, 2) - with the
    generator seeded, exactly the ranking those draws give; any other name is refused."""
    _, results = data
    r = _first_rich(results)
    sRT = r["multi_hypothesis_pred_RTs"]
    n, K = sRT.shape[:2]
    np.random.seed(5)
    sel, avg, sel_e = ev.sort_sRT_by_energy(sRT.copy(), None, None, "random", 0.6, "average")
    np.random.seed(5)
    draws = np.random.rand(n, K, 2)
    want, _, want_e = ev.sort_sRT_by_energy(sRT.copy(), draws, None, "energy_ranker", 0.6, "nearest")
    np.testing.assert_array_equal(sel, want)
    np.testing.assert_array_equal(sel_e, want_e)
    assert avg.shape == (n, 4, 4) and not np.array_equal(sel, ev.sort_sRT_by_energy(sRT.copy(), r["energy"].copy(), None, "energy_ranker", 0.6, "nearest")[0])
    with pytest.raises(NotImplementedError):
        ev.sort_sRT_by_energy(sRT.copy(), r["energy"], None, "bogus", 0.6, "average")


@pytest.mark.parametrize("mode", ["average", "nearest"])
def test_compute_mAP_matches_reference(data, mode, tmp_path):
    g, results = data
    iou_aps, pose_aps, iou_acc, pose_acc = ev.compute_mAP(copy.deepcopy(results), str(tmp_path), DEGREE, SHIFT, IOU, iou_pose_thres=0.1,
                                                           use_matches_for_pose=True, repeat_num=int(g["K"]), pooling_mode=mode, ratio=0.6,
                                                           ranker="energy_ranker")
    for got, name in ((iou_aps, "iou_aps"), (pose_aps, "pose_aps"), (iou_acc, "iou_acc"), (pose_acc, "pose_acc")):
        ref = g[f"{mode}_{name}"]
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12, equal_nan=True, err_msg=name)
    assert (tmp_path / "mAP_Acc.pkl").exists()
    s = ev.summary(iou_aps, pose_aps, IOU, DEGREE + [360], SHIFT + [100])
    assert set(s) >= {"IoU25", "IoU50", "IoU75", "5deg2cm", "5deg5cm", "10deg2cm", "10deg5cm"}
    assert 0 < s["10deg5cm"] <= 100


def test_symmetry_rules():
    eye = np.eye(4)
    spin = np.eye(4)
    c, s_ = np.cos(1.0), np.sin(1.0)
    spin[:3, :3] = [[c, 0, s_], [0, 1, 0], [-s_, 0, c]]  # 57.3 deg about y
    for cls, hv, expect_zero in ((1, 1, True), (2, 1, True), (4, 1, True), (3, 1, False), (5, 1, False), (6, 1, False), (6, 0, True)):
        rt = ev.compute_RT_overlaps(np.array([cls]), eye[None], np.array([hv]), np.array([cls]), spin[None])
        assert (rt[0, 0, 0] < 1e-6) == expect_zero, (cls, hv, rt)
        if not expect_zero:
            assert abs(rt[0, 0, 0] - np.degrees(1.0)) < 1e-9
    scaled = eye.copy()
    scaled[:3, :3] *= 0.3  # NOCS sRT carries the scale: it must not change the angle
    shifted = spin.copy()
    shifted[:3, 3] = [0.03, 0.0, 0.04]
    rt = ev.compute_RT_overlaps(np.array([3]), scaled[None], np.array([1]), np.array([3]), shifted[None])
    assert abs(rt[0, 0, 0] - np.degrees(1.0)) < 1e-9 and abs(rt[0, 0, 1] - 5.0) < 1e-12


def test_empty_and_ragged_inputs():
    assert ev.compute_RT_overlaps(np.zeros(0), np.zeros((0, 4, 4)), np.zeros(0), np.zeros(0), None).shape == (0, 0, 2)
    gm, pm = ev.compute_RT_matches(np.zeros((0, 2, 2)), np.zeros(0), np.array([1, 1]), [5], [5])
    assert gm.shape == (1, 1, 2) and pm.shape == (1, 1, 0) and (gm == -1).all()
    gm, pm, ov, idx = ev.compute_2d_IoU_matches(np.array([1]), np.array([[0, 0, 10, 10]]), np.zeros(0), np.zeros((0, 4)), np.zeros(0), [0.1, 0.5])
    assert gm.shape == (2, 1) and pm.shape == (2, 0) and len(idx) == 0
    with np.errstate(all="ignore"):
        ap, acc = ev.compute_ap_and_acc(np.zeros(0), np.zeros(0), np.array([-1.0]))
    assert ap == 0 and np.isnan(acc)
    # an image without objects or detections is skipped, a class without data yields nan accuracy like the reference
    res = [{"gt_class_ids": np.zeros(0, dtype=np.int32), "gt_RTs": np.zeros((0, 4, 4)), "gt_scales": np.zeros((0, 3)), "gt_bboxes": np.zeros((0, 4)),
            "gt_handle_visibility": np.zeros(0), "pred_class_ids": np.zeros(0, dtype=np.int32), "pred_bboxes": np.zeros((0, 4)),
            "pred_scores": np.zeros(0), "pred_scales": np.zeros((0, 3)), "multi_hypothesis_pred_RTs": np.zeros((0, 4, 4, 4)), "energy": np.zeros((0, 4, 2))}]
    iou_aps, pose_aps, iou_acc, pose_acc = ev.compute_mAP(res, None, [5], [5], [0.1, 0.5], iou_pose_thres=0.1, use_matches_for_pose=True, repeat_num=4)
    assert iou_aps.shape == (8, 2) and pose_aps.shape == (8, 2, 2) and np.isnan(iou_acc[1:7]).all()


def test_detection_results_container():
    K = 3
    det = {"img_a": {"result": {"pred_RTs": np.tile(np.eye(4), (2, 1, 1)), "gt_class_ids": np.array([1, 6])},
                     "valid_pts": [np.zeros((1024, 3), dtype=np.float32)], "valid_rgb": None, "cat_id": [5], "valid_inst": [1]},
           "img_b": {"result": {"pred_RTs": np.tile(np.eye(4), (1, 1, 1)), "gt_class_ids": np.array([2])},
                     "valid_pts": [np.ones((1024, 3), dtype=np.float32)], "valid_rgb": None, "cat_id": [5], "valid_inst": [0]}}
    store = ev.DetectionResults(det, K)
    assert det["img_a"]["result"]["multi_hypothesis_pred_RTs"].shape == (2, K, 4, 4) and det["img_a"]["result"]["energy"].shape == (2, K, 2)
    (sl, pts), = list(store.batches("mug", 8))
    assert pts.shape == (2, 1024, 3) and pts.dtype == np.float32
    RT = np.tile(np.eye(4), (2, K, 1, 1))
    RT[0, :, 0, 3] = 7.0
    RT[1, :, 0, 3] = 9.0
    store.write("mug", sl, RT, np.full((2, K, 2), 0.5))
    assert det["img_a"]["result"]["multi_hypothesis_pred_RTs"][1, 0, 0, 3] == 7.0   # instance 1 of img_a
    assert det["img_a"]["result"]["multi_hypothesis_pred_RTs"][0, 0, 0, 3] == 0.0   # invalid instance keeps identity
    assert det["img_b"]["result"]["multi_hypothesis_pred_RTs"][0, 0, 0, 3] == 9.0 and det["img_b"]["result"]["energy"][0, 0, 0] == 0.5
    assert all("gt_handle_visibility" in r for r in store.results())
