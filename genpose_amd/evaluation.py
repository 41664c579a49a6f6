"""Result containers and NOCS-style mAP evaluation - the step right after the hot path.

Host-side numpy by design (per-image work is a handful of instances; the reference's cost is Python loops over
46 x 22 pose thresholds and 101 IoU thresholds per image and class, vectorised here over the threshold axes).
Same inputs, same outputs and the same matching / AP rules as the reference:

  unpack_data / write-back     runners/evaluation_single.py:263-306, 411-416, 474-481   -> DetectionResults
  compute_2d_IoU_matches       utils/sgpa_utils.py:468-526
  compute_RT_errors/_overlaps  utils/sgpa_utils.py:530-586   (y-axis symmetry for bottle / can / bowl and handle-less mugs)
  compute_RT_matches           utils/sgpa_utils.py:589-624
  compute_ap_and_acc           utils/sgpa_utils.py:627-650
  sort_sRT, sort_sRT_by_energy utils/sgpa_utils.py:24-51, 897-954  (energy ranker; rotation and translation ranked independently)
  compute_mAP                  utils/sgpa_utils.py:957-1183  (use_matches_for_pose=True, 2-D IoU matching as evaluate() calls it)

Pinned by tests/golden/g10_map.npz, produced by the imported reference (oracle/gen_golden.py).
"""
import os
import pickle

import numpy as np
import torch

from . import rotation

SYNSET_NAMES = ["BG", "bottle", "bowl", "camera", "can", "laptop", "mug"]
_Y_SYMMETRIC = ("bottle", "can", "bowl")


# ---------------------------------------------------------------------------------------------- containers
class DetectionResults:
    """The reference's `detect_result` dict (img_path -> {'result', 'valid_pts', 'cat_id', 'valid_inst'}) plus the per-category
    flattening `unpack_data` builds, with the write-back of hypotheses / energies by (img_path, instance)."""

    def __init__(self, detect_result, repeat_num, synset_names=("bottle", "bowl", "camera", "can", "laptop", "mug")):
        self.detect_result = detect_result
        self.repeat_num = repeat_num
        self.by_category = {c: {"img_path": [], "pts": [], "cat_id": [], "inst": []} for c in synset_names}
        for key, entry in detect_result.items():
            res = entry["result"]
            n = res["pred_RTs"].shape[0]
            res["multi_hypothesis_pred_RTs"] = np.tile(np.identity(4, dtype=float), (n, repeat_num, 1, 1))
            res["energy"] = np.zeros((n, repeat_num, 2), dtype=float)
            for i, inst in enumerate(entry["valid_inst"]):
                cat = synset_names[entry["cat_id"][i]]
                d = self.by_category[cat]
                d["img_path"].append(key)
                d["pts"].append(entry["valid_pts"][i])
                d["cat_id"].append(entry["cat_id"][i])
                d["inst"].append(inst)

    def batches(self, category, batch_size):
        """Yields (slice, pts [b,1024,3] float32) over a category, the reference's batching (evaluation_single.py:380-382)."""
        d = self.by_category[category]
        n = len(d["img_path"])
        for s in range(0, n, batch_size):
            sl = slice(s, min(n, s + batch_size))
            yield sl, np.asarray(d["pts"][sl], dtype=np.float32)

    def write(self, category, sl, pred_RTs, energy=None):
        """pred_RTs [b,K,4,4] (+ energy [b,K,2]) of the instances `sl` of `category` -> detect_result (…:414-416, 476-479)."""
        d = self.by_category[category]
        for k, (path, inst) in enumerate(zip(d["img_path"][sl], d["inst"][sl])):
            res = self.detect_result[path]["result"]
            res["multi_hypothesis_pred_RTs"][inst] = np.asarray(pred_RTs[k])
            if energy is not None:
                res["energy"][inst] = np.asarray(energy[k])

    def results(self):
        """List of per-image result dicts as `evaluate` hands them to compute_mAP (…:505-518)."""
        out = []
        for entry in self.detect_result.values():
            r = entry["result"]
            if "gt_handle_visibility" not in r:
                r["gt_handle_visibility"] = np.ones_like(r["gt_class_ids"])
            out.append(r)
        return out


# ---------------------------------------------------------------------------------------------- ranking / aggregation
def sort_sRT(poses, energy):
    """poses [n,K,4,4], energy [n,K,2] -> hypotheses ranked from highest to lowest energy, rotation (and the rest of the
    matrix) by energy[...,0], the translation column by energy[...,1]."""
    order = np.argsort(-energy, axis=1)
    sorted_energy = -np.sort(-energy, axis=1)
    rows = np.arange(poses.shape[0])[:, None]
    out = poses[rows, order[:, :, 0]].copy()
    out[:, :, :, 3] = poses[rows, order[:, :, 1]][:, :, :, 3]
    return out, sorted_energy


def average_sRT(selected):
    """[n,m,4,4] -> [n,4,4]: quaternion average (largest eigenvector of the mean outer product) + mean translation."""
    n, m = selected.shape[:2]
    s = torch.from_numpy(np.ascontiguousarray(selected)).reshape(n * m, 4, 4)
    q = rotation.matrix_to_quaternion(s[:, :3, :3])
    qt = torch.cat((q, s[:, :3, 3]), dim=-1).reshape(n, m, -1)
    avg = torch.zeros((n, 7))  # float32, as the reference's torch.zeros
    avg[:, :4] = rotation.average_quaternion_batch(qt[:, :, :4])  # float64 in, stored as float32 (sgpa_utils.py:943-945)
    avg[:, 4:] = torch.mean(qt[:, :, 4:], dim=1)
    out = np.tile(np.identity(4), (n, 1, 1))
    out[:, :3, :3] = rotation.quaternion_to_matrix(avg[:, :4]).numpy()
    out[:, :3, 3] = avg[:, 4:].numpy()
    return out


def sort_sRT_by_energy(sRT, energy=None, RT_overlaps=None, ranker="energy_ranker", ratio=1.0, error_mode="average"):
    n, K = sRT.shape[:2]
    m = max(1, int(K * ratio))
    if n == 0:
        return sRT[:, :m], None, None
    if ranker == "energy_ranker":
        e = energy
    elif ranker == "gt_ranker":
        e = -np.min(RT_overlaps, axis=1)
    elif ranker == "random":
        e = np.random.rand(n, K, 2)  # sgpa_utils.py:926-927: numpy's global generator (seed it for a repeatable ablation)
    else:
        raise NotImplementedError(ranker)
    s, se = sort_sRT(sRT, e)
    s, se = s[:, :m], se[:, :m]
    if error_mode == "average":
        return s, average_sRT(s), se
    if error_mode == "nearest":
        return s, None, se
    raise NotImplementedError(error_mode)


# ---------------------------------------------------------------------------------------------- overlaps
def compute_2d_IoU(a, b):
    """Pairwise IoU of inclusive integer boxes [x1,y1,x2,y2]: a [P,4], b [G,4] -> [P,G] float64."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4)[:, None, :]
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4)[None, :, :]
    iw = np.maximum(0, np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0]) + 1)
    ih = np.maximum(0, np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1]) + 1)
    inter = iw * ih
    area_a = (a[..., 2] - a[..., 0] + 1) * (a[..., 3] - a[..., 1] + 1)
    area_b = (b[..., 2] - b[..., 0] + 1) * (b[..., 3] - b[..., 1] + 1)
    return inter / (area_a + area_b - inter)


def compute_RT_overlaps(gt_class_ids, gt_sRT, gt_handle_visibility, pred_class_ids, pred_sRT, synset_names=SYNSET_NAMES):
    """[P,G,2]: rotation error in degrees and translation error in centimetres of every (prediction, ground truth) pair;
    the symmetry rule is the GROUND TRUTH's class (and handle visibility for mugs)."""
    P, G = len(pred_class_ids), len(gt_class_ids)
    out = np.zeros((P, G, 2))
    if P == 0 or G == 0:
        return out
    pred_sRT, gt_sRT = np.asarray(pred_sRT, dtype=np.float64), np.asarray(gt_sRT, dtype=np.float64)
    last = np.array([0.0, 0.0, 0.0, 1.0])
    if not (np.array_equal(pred_sRT[:, 3, :], np.tile(last, (P, 1))) and np.array_equal(gt_sRT[:, 3, :], np.tile(last, (G, 1)))):
        raise ValueError("sRT matrices must end in the row [0, 0, 0, 1]")
    R1 = pred_sRT[:, :3, :3] / np.cbrt(np.linalg.det(pred_sRT[:, :3, :3]))[:, None, None]
    R2 = gt_sRT[:, :3, :3] / np.cbrt(np.linalg.det(gt_sRT[:, :3, :3]))[:, None, None]
    names = [synset_names[int(c)] for c in gt_class_ids]
    sym = np.array([(nm in _Y_SYMMETRIC) or (nm == "mug" and gt_handle_visibility[j] == 0) for j, nm in enumerate(names)])
    # general case: cos = (trace(R1 R2^T) - 1) / 2
    full = np.einsum("pik,gjk->pgij", R1, R2)
    cos_full = (np.trace(full, axis1=2, axis2=3) - 1) / 2
    # symmetric about y: angle between the two y axes
    y1, y2 = R1[:, :, 1], R2[:, :, 1]
    cos_sym = (y1 @ y2.T) / (np.linalg.norm(y1, axis=1)[:, None] * np.linalg.norm(y2, axis=1)[None, :])
    cos = np.where(sym[None, :], cos_sym, cos_full)
    out[:, :, 0] = np.arccos(np.clip(cos, -1.0, 1.0)) * 180 / np.pi
    out[:, :, 1] = np.linalg.norm(pred_sRT[:, None, :3, 3] - gt_sRT[None, :, :3, 3], axis=-1) * 100
    return out


# ---------------------------------------------------------------------------------------------- matching
def compute_2d_IoU_matches(gt_class_ids, gt_bboxes, pred_class_ids, pred_bboxes, pred_scores, iou_2d_thresholds, score_threshold=0):
    """Greedy matching of predictions (in descending score order) to ground truth per IoU threshold.
    Returns gt_matches [T,G], pred_matches [T,P] (indices, -1 = none), overlaps [P,G] float32, indices (score order)."""
    P, G = len(pred_class_ids), len(gt_class_ids)
    thr = np.asarray(iou_2d_thresholds, dtype=np.float64)
    T = len(thr)
    indices = np.zeros(0)
    if P:
        indices = np.argsort(pred_scores)[::-1]
        pred_class_ids = pred_class_ids[indices].copy()
    # NB (as in the reference): the overlaps stay in the callers' prediction order while the classes are re-ordered by score
    overlaps = compute_2d_IoU(pred_bboxes, gt_bboxes).astype(np.float32) if P and G else np.zeros((P, G), dtype=np.float32)
    pred_matches = -1 * np.ones([T, P])
    gt_matches = -1 * np.ones([T, G])
    for i in range(len(indices)):
        order = np.argsort(overlaps[i])[::-1]
        low = np.where(overlaps[i, order] < score_threshold)[0]
        if low.size > 0:
            order = order[: low[0]]
        active = np.ones(T, dtype=bool)  # thresholds for which prediction i is still looking
        for j in order:
            iou = overlaps[i, j]
            free = active & (gt_matches[:, j] <= -1)
            active &= ~(free & (iou < thr))          # below the threshold: stop for this threshold
            free &= active
            if pred_class_ids[i] != gt_class_ids[j]:
                continue
            hit = free & (iou > thr)
            gt_matches[hit, j] = i
            pred_matches[hit, i] = j
            active &= ~hit
            if not active.any():
                break
    return gt_matches, pred_matches, overlaps, indices


def compute_RT_matches(overlaps, pred_class_ids, gt_class_ids, degree_thres_list, shift_thres_list):
    """Greedy pose matching per (degree, shift) threshold pair: predictions in order, candidates by ascending
    (rotation error + translation error).  Returns gt_matches [D,S,G], pred_matches [D,S,P]."""
    D, S = len(degree_thres_list), len(shift_thres_list)
    P, G = len(pred_class_ids), len(gt_class_ids)
    pred_matches = -1 * np.ones((D, S, P))
    gt_matches = -1 * np.ones((D, S, G))
    if P == 0 or G == 0:
        return gt_matches, pred_matches
    assert overlaps.shape == (P, G, 2)
    deg = np.asarray(degree_thres_list, dtype=np.float64)[:, None]
    sh = np.asarray(shift_thres_list, dtype=np.float64)[None, :]
    for i in range(P):
        order = np.argsort(np.sum(overlaps[i], axis=-1))
        open_ = np.ones((D, S), dtype=bool)  # threshold pairs for which prediction i is unmatched so far
        for j in order:
            if pred_class_ids[i] != gt_class_ids[j]:
                continue
            ok = open_ & (gt_matches[:, :, j] <= -1) & ~((overlaps[i, j, 0] > deg) | (overlaps[i, j, 1] > sh))
            gt_matches[ok, j] = i
            pred_matches[ok, i] = j
            open_ &= ~ok
            if not open_.any():
                break
    return gt_matches, pred_matches


def compute_ap_and_acc(pred_matches, pred_scores, gt_matches):
    """VOC-style AP and accuracy.  Accepts a leading batch of threshold axes: pred_matches [...,N], pred_scores [...,N]
    (identical along the batch axes, as compute_mAP tiles them), gt_matches [...,M] -> ap [...], acc [...]."""
    pred_matches, pred_scores, gt_matches = np.asarray(pred_matches), np.asarray(pred_scores), np.asarray(gt_matches)
    assert pred_matches.shape == pred_scores.shape
    lead = pred_matches.shape[:-1]
    N, M = pred_matches.shape[-1], gt_matches.shape[-1]
    nb = int(np.prod(lead)) if lead else 1
    pm = pred_matches.reshape(nb, N)
    ps = pred_scores.reshape(nb, N)
    ap = np.zeros(pm.shape[0])
    acc = np.zeros(pm.shape[0])
    with np.errstate(divide="ignore", invalid="ignore"):
        for t in range(pm.shape[0]):
            idx = np.argsort(ps[t])[::-1]
            hit = pm[t][idx] > -1
            csum = np.cumsum(hit)
            precisions = csum / (np.arange(N) + 1)
            recalls = csum.astype(np.float32) / M
            precisions = np.concatenate([[0], precisions, [0]])
            recalls = np.concatenate([[0], recalls, [1]])
            precisions = np.maximum.accumulate(precisions[::-1])[::-1]  # non-increasing envelope
            k = np.where(recalls[:-1] != recalls[1:])[0] + 1
            ap[t] = np.sum((recalls[k] - recalls[k - 1]) * precisions[k])
            acc[t] = np.sum(hit) / N
    return (ap.reshape(lead), acc.reshape(lead)) if lead else (ap[0], acc[0])


# ---------------------------------------------------------------------------------------------- mAP
def compute_mAP(pred_results, out_dir=None, degree_thresholds=(180,), shift_thresholds=(100,), iou_3d_thresholds=(0.1,), iou_pose_thres=0.1,
                use_matches_for_pose=False, repeat_num=None, pooling_mode="average", ratio=1.0, so3_vis=False, ranker="energy_ranker",
                synset_names=SYNSET_NAMES):
    """Mean average precision / accuracy over IoU thresholds and (degree, shift) threshold pairs.

    pred_results: list of per-image dicts with gt_class_ids, gt_RTs, gt_scales, gt_handle_visibility, gt_bboxes,
    pred_class_ids, pred_bboxes, pred_scores, multi_hypothesis_pred_RTs [n,K,4,4], energy [n,K,2] (DetectionResults.results()).
    Returns iou_aps [C+1,T], pose_aps [C+1,D+1,S+1], iou_acc, pose_acc (last row = mean over the object classes; the extra
    degree / shift entries are the 360 deg / 100 cm catch-alls).  Writes mAP_Acc.pkl into out_dir when given."""
    if so3_vis:
        raise NotImplementedError("so3_vis (interactive SO(3) plots) is not part of the evaluation path")
    C = len(synset_names)
    deg_list = list(degree_thresholds) + [360]
    sh_list = list(shift_thresholds) + [100]
    iou_list = list(iou_3d_thresholds)
    D, S, T = len(deg_list), len(sh_list), len(iou_list)
    if use_matches_for_pose:
        assert iou_pose_thres in iou_list
        thres_ind = iou_list.index(iou_pose_thres)
    iou_pm = [[] for _ in range(C)]
    iou_ps = [[] for _ in range(C)]
    iou_gm = [[] for _ in range(C)]
    pose_pm = [[] for _ in range(C)]
    pose_ps = [[] for _ in range(C)]
    pose_gm = [[] for _ in range(C)]

    for result in pred_results:
        gt_class_ids = result["gt_class_ids"].astype(np.int32)
        gt_sRT = np.array(result["gt_RTs"])
        gt_hv = result["gt_handle_visibility"]
        gt_bboxes = np.array(result["gt_bboxes"])
        pred_class_ids = result["pred_class_ids"]
        hyp = result["multi_hypothesis_pred_RTs"]
        energy = result["energy"]
        pred_scores = result["pred_scores"]
        pred_bboxes = np.array(result["pred_bboxes"])
        if len(gt_class_ids) == 0 and len(pred_class_ids) == 0:
            continue
        K = repeat_num if repeat_num is not None else (hyp.shape[1] if len(pred_class_ids) else 0)
        for c in range(1, C):
            gsel = gt_class_ids == c if len(gt_class_ids) else np.zeros(0, dtype=bool)
            psel = pred_class_ids == c if len(pred_class_ids) else np.zeros(0, dtype=bool)
            c_gt_ids = gt_class_ids[gsel] if len(gt_class_ids) else np.zeros(0)
            c_gt_sRT = gt_sRT[gsel] if len(gt_class_ids) else np.zeros((0, 4, 4))
            c_gt_boxes = gt_bboxes[gsel] if len(gt_class_ids) else np.zeros((0, 4))
            if synset_names[c] != "mug":
                c_gt_hv = np.ones_like(c_gt_ids)
            else:
                c_gt_hv = gt_hv[gsel] if len(gt_class_ids) else np.ones(0)
            c_pred_ids = pred_class_ids[psel] if len(pred_class_ids) else np.zeros(0)
            c_hyp = hyp[psel] if len(pred_class_ids) else np.zeros((0, K, 4, 4))
            c_energy = energy[psel] if len(pred_class_ids) else np.zeros((0, K, 2))
            c_scores = pred_scores[psel] if len(pred_class_ids) else np.zeros(0)
            c_boxes = pred_bboxes[psel] if len(pred_class_ids) else np.zeros((0, 4))

            gm, pm, _, order = compute_2d_IoU_matches(c_gt_ids, c_gt_boxes, c_pred_ids, c_boxes, c_scores, iou_list)
            if len(order):
                c_pred_ids, c_hyp, c_energy, c_scores = c_pred_ids[order], c_hyp[order], c_energy[order], c_scores[order]
            iou_pm[c].append(pm)
            iou_ps[c].append(np.tile(c_scores, (T, 1)))
            iou_gm[c].append(gm)
            if use_matches_for_pose:
                keep_p = pm[thres_ind] > -1
                keep_g = gm[thres_ind] > -1
                c_pred_ids = c_pred_ids[keep_p] if len(keep_p) else np.zeros(0)
                c_hyp = c_hyp[keep_p] if len(keep_p) else np.zeros((0, K, 4, 4))
                c_energy = c_energy[keep_p] if len(keep_p) else np.zeros((0, K, 2))
                c_scores = c_scores[keep_p] if len(keep_p) else np.zeros(0)
                c_gt_ids = c_gt_ids[keep_g] if len(keep_g) else np.zeros(0)
                c_gt_sRT = c_gt_sRT[keep_g] if len(keep_g) else np.zeros((0, 4, 4))
                c_gt_hv = c_gt_hv[keep_g] if len(keep_g) else np.zeros(0)

            hyp_overlaps = None
            if ranker == "gt_ranker":
                per_h = [compute_RT_overlaps(c_gt_ids, c_gt_sRT, c_gt_hv, c_pred_ids, c_hyp[:, i], synset_names) for i in range(c_hyp.shape[1])]
                hyp_overlaps = np.array(per_h).transpose(1, 2, 0, 3)
            if pooling_mode == "nearest":
                s_hyp, _, _ = sort_sRT_by_energy(c_hyp, c_energy, hyp_overlaps, ranker, ratio, "nearest")
                per_h = [compute_RT_overlaps(c_gt_ids, c_gt_sRT, c_gt_hv, c_pred_ids, s_hyp[:, i], synset_names) for i in range(s_hyp.shape[1])]
                RT_overlaps = np.min(np.array(per_h), axis=0)
            elif pooling_mode == "average":
                _, avg, _ = sort_sRT_by_energy(c_hyp, c_energy, hyp_overlaps, ranker, ratio, "average")
                RT_overlaps = compute_RT_overlaps(c_gt_ids, c_gt_sRT, c_gt_hv, c_pred_ids, avg, synset_names)
            else:
                raise NotImplementedError(pooling_mode)
            pgm, ppm = compute_RT_matches(RT_overlaps, c_pred_ids, c_gt_ids, deg_list, sh_list)
            pose_pm[c].append(ppm)
            pose_ps[c].append(np.tile(c_scores, (D, S, 1)))
            pose_gm[c].append(pgm)

    cat = lambda parts, shape: np.concatenate(parts, axis=-1) if parts else np.zeros(shape)
    iou_aps, iou_acc = np.zeros((C + 1, T)), np.zeros((C + 1, T))
    pose_aps, pose_acc = np.zeros((C + 1, D, S)), np.zeros((C + 1, D, S))
    for c in range(1, C):
        iou_aps[c], iou_acc[c] = compute_ap_and_acc(cat(iou_pm[c], (T, 0)), cat(iou_ps[c], (T, 0)), cat(iou_gm[c], (T, 0)))
        pose_aps[c], pose_acc[c] = compute_ap_and_acc(cat(pose_pm[c], (D, S, 0)), cat(pose_ps[c], (D, S, 0)), cat(pose_gm[c], (D, S, 0)))
    iou_aps[-1], iou_acc[-1] = np.mean(iou_aps[1:-1], axis=0), np.mean(iou_acc[1:-1], axis=0)
    pose_aps[-1], pose_acc[-1] = np.mean(pose_aps[1:-1], axis=0), np.mean(pose_acc[1:-1], axis=0)
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "mAP_Acc.pkl"), "wb") as f:
            pickle.dump({"iou_thres_list": iou_list, "degree_thres_list": deg_list, "shift_thres_list": sh_list, "iou_aps": iou_aps,
                         "pose_aps": pose_aps, "iou_acc": iou_acc, "pose_acc": pose_acc}, f)
    return iou_aps, pose_aps, iou_acc, pose_acc


def summary(iou_aps, pose_aps, iou_thres_list, degree_thres_list, shift_thres_list):
    """The headline numbers the reference logs (evaluation_single.py: record_results_and_draw_curves): mean over classes of
    IoU25/50/75 and 5deg2cm, 5deg5cm, 10deg2cm, 10deg5cm (percent)."""
    it, dt, st = list(iou_thres_list), list(degree_thres_list), list(shift_thres_list)
    out = {}
    for name, v in (("IoU25", 0.25), ("IoU50", 0.5), ("IoU75", 0.75)):
        if v in it:
            out[name] = 100 * iou_aps[-1, it.index(v)]
    for d, s in ((5, 2), (5, 5), (10, 2), (10, 5), (10, 10)):
        if d in dt and s in st:
            out[f"{d}deg{s}cm"] = 100 * pose_aps[-1, dt.index(d), st.index(s)]
    return out
