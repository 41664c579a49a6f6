"""ORACLE — TEST INFRASTRUCTURE ONLY; runs ONLY in the build container (needs /root/reference).

Generates tests/golden/*.npz by RUNNING THE IMPORTED REFERENCE (oracle/ref_import.py) on seeded inputs:

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

Fixtures hold data only (inputs, logged random draws, expected outputs).  Weights are NOT stored: they
are regenerated from oracle.genpose_oracle.make_state_dict(seed, mode) (reference key names; loaded into
the reference nets with load_state_dict(strict=True), which also pins the key schema / shapes).

What each file pins:
  g1_g2_ops.npz   FPS idx (3 levels) and ball-query idx (6 calls) through the reference's own autograd
                  Functions (pointnet2_utils.py) - backed by the C restatement, see pn2_ops.c header.
  g3_encoder.npz  Pointnet2ClsMSG(0) forward, BN stats randomised.
  g4_g5_nets.npz  PoseScoreNet / PoseEnergyNet at t in {1e-5, 0.15, 0.55, 1.0}.
  g6_ode.npz      PoseNet.pred_func with the ODE sampler (T0 1.0 / 0.55, sampling_steps None / 20, warm start).
  g7_pc.npz       PoseNet.pred_func with the PC sampler, 20 steps, logged randn_like draws.
  g8_rank.npz     get_energy + sort_poses_by_energy + sort_sRT_by_energy(ratio=.6,'average').
  g9_track.npz    3-frame tracking loop semantics (evaluation_tracking.py:262-337) incl. add_noise_to_RT draws.
  g10..g13        mAP evaluation, depth -> cloud pre-processing, likelihood, score of the energy model (--g10 .. --g13).
  g14_train_step.npz  one training step of the score model (--g14): loss, clipped gradients, Adam update, EMA, BN statistics.
  g15_energy_train_step.npz  one training step of the energy model incl. the ranking loss (--g15).
  g16_encoder_{dense,lighter}.npz  the encoder under the reference's other configurations (--g16 dense | --g16 lighter; one process each).
"""
import hashlib
import os
import sys

import numpy as np
import torch

from . import genpose_oracle as go
from . import ref_import
from genpose_amd import synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class DrawLog:
    """Logs every torch.randn / torch.randn_like result while active (call order preserved)."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like
        log = self.draws

        def randn(*a, **k):
            r = self._randn(*a, **k)
            log.append(r.detach().clone())
            return r

        def randn_like(x, **k):
            r = self._randn_like(x, **k)
            log.append(r.detach().clone())
            return r

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def make_agent(ns, mode, seed=0, sampler="ode", sampling_steps=None):
    import argparse
    cfg = argparse.Namespace(**vars(ns.cfg))
    cfg.posenet_mode = mode
    cfg.sampler_mode = [sampler]
    cfg.sampling_steps = sampling_steps
    agent = ns.PoseNet(cfg)
    sd = go.make_state_dict(seed, mode)
    agent.net.load_state_dict(sd, strict=True)
    agent.net.eval()
    return agent, sd


def score_time_log(agent):
    """Wrap the score net so every evaluation's t is recorded."""
    times = []
    net = agent.net.pose_score_net
    orig = net.forward

    def fwd(data, *a, **k):
        times.append(float(data["t"][0, 0]))
        return orig(data, *a, **k)

    net.forward = fwd
    return times, lambda: setattr(net, "forward", orig)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gen_g1_g2(ns, arith=None):
    """G1/G2 through the reference's own autograd Functions (pointnet2_utils.py) with the stand-in module under the arithmetic convention
    `arith` (oracle/pn2_ops.c header; None = the oracle's default).  The default convention's file is g1_g2_ops.npz, the others
    g1_g2_ops_arith<X>.npz - small files, one per convention, so that every arm of the switch is held to a reference-driven vector."""
    from . import pn2_oracle
    arith = arith or pn2_oracle.DEFAULT_ARITH
    P = ns.pn2_utils
    with pn2_oracle.use_arith(arith):
        clouds = synth.golden_clouds()  # [4,1024,3]: 2 surface-like, 1 tiled-duplicate, 1 grid ties
        xyz = torch.from_numpy(clouds)
        g = {"clouds": clouds, "arith": np.array(arith)}
        cur = xyz
        for lvl, (npnt, radii, nss) in enumerate(zip([512, 256, 128], go.LIGHT_CFG["radii"], go.LIGHT_CFG["nsamples"])):
            idx = P.furthest_point_sample(cur.contiguous(), npnt)
            new = P.gather_operation(cur.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
            g[f"fps_idx{lvl}"] = idx.numpy().astype(np.int16)
            for s, (r, nsmp) in enumerate(zip(radii, nss)):
                bq = P.ball_query(r, nsmp, cur.contiguous(), new).numpy()
                g[f"bq{lvl}_{s}_cloud0"] = bq[0].astype(np.int16)
                g[f"bq{lvl}_{s}_sha"] = np.array([sha(bq[b].astype(np.int32)) for b in range(bq.shape[0])])
            cur = new
        # odd sizes: n not a power of two, tiny nsample
        odd = torch.from_numpy(synth.golden_clouds(seed=77)[:2, :700].copy())
        g["odd_clouds"] = odd.numpy()
        g["odd_fps"] = P.furthest_point_sample(odd.contiguous(), 100).numpy().astype(np.int16)
        g["odd_bq"] = P.ball_query(0.05, 5, odd.contiguous(), odd[:, :50].contiguous()).numpy().astype(np.int16)
        # three_nn / three_interpolate (north_star lists them; the encoder does not reach them): the other two sites of the contraction
        unknown, known = xyz[:2, :300].contiguous(), xyz[:2, 300:364].contiguous()
        # ThreeNN.forward returns torch.sqrt(dist2) (pointnet2_utils.py:99) and a vectorised host sqrt is not the same bits on every CPU:
        # the fixture keeps dist2 itself, captured with torch.sqrt stubbed to the identity for this one call
        _sqrt, torch.sqrt = torch.sqrt, (lambda t: t)
        try:
            d2, i3 = P.three_nn(unknown, known)
        finally:
            torch.sqrt = _sqrt
        w = 1.0 / (d2.double().sqrt() + 1e-8)
        w = (w / w.sum(dim=2, keepdim=True)).float().contiguous()
        feats = torch.from_numpy(np.random.default_rng(5).standard_normal((2, 7, 64)).astype(np.float32))
        g["nn_dist2"], g["nn_idx"] = d2.numpy(), i3.numpy().astype(np.int16)
        g["interp_feats"], g["interp_w"] = feats.numpy(), w.numpy()
        g["interp_out"] = P.three_interpolate(feats, i3, w).numpy()
    name = "g1_g2_ops.npz" if arith == pn2_oracle.DEFAULT_ARITH else f"g1_g2_ops_arith{arith}.npz"
    np.savez_compressed(os.path.join(OUT, name), **g)
    return clouds, xyz


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = ref_import.load()
    torch.set_grad_enabled(False)
    P = ns.pn2_utils

    clouds, xyz = gen_g1_g2(ns)

    # ---------------- G3 encoder
    agent, sd = make_agent(ns, "score")
    feat = agent.net({"pts": xyz}, mode="pts_feature")
    # per-level features for cloud 0 through forward hooks on the SA modules
    inter = []
    hooks = [m.register_forward_hook(lambda m, i, o: inter.append(o)) for m in agent.net.pts_encoder.SA_modules]
    agent.net({"pts": xyz[:1]}, mode="pts_feature")
    for h in hooks:
        h.remove()
    g3 = {"clouds": clouds, "feat": feat.numpy(), "seed": np.array(0)}
    for lvl in range(3):
        nx, f = inter[lvl]
        g3[f"new_xyz{lvl}"] = nx[0].numpy()
        g3[f"feat{lvl}_first32"] = f[0, :, :32].numpy()  # [C, 32 points]
    np.savez_compressed(os.path.join(OUT, "g3_encoder.npz"), **g3)

    # ---------------- G4/G5 nets
    gen = torch.Generator().manual_seed(11)
    pf = torch.randn(8, 1024, generator=gen).abs()
    pose = torch.randn(8, 9, generator=gen)
    g4 = {"pts_feat": pf.numpy(), "pose": pose.numpy(), "t": np.array([1e-5, 0.15, 0.55, 1.0], dtype=np.float32)}
    e_agent, sd_e = make_agent(ns, "energy")
    for i, t in enumerate(g4["t"]):
        tt = torch.ones(8, 1) * float(t)
        data = {"pts_feat": pf, "sampled_pose": pose, "t": tt}
        g4[f"score_{i}"] = agent.net(data, mode="score").numpy()
        g4[f"energy_{i}"] = e_agent.net(data, mode="energy").numpy()
    np.savez_compressed(os.path.join(OUT, "g4_g5_nets.npz"), **g4)

    # ---------------- G6 ODE sampler (B=2, K=10)
    pts2 = xyz[:2].clone()
    cen2 = pts2.mean(dim=1)
    g6 = {"pts": pts2.numpy(), "K": np.array(10)}
    cases = [("T1_none", 1.0, None, False), ("T055_none", 0.55, None, False), ("T055_s20", 0.55, 20, False),
             ("T015_warm", 0.15, None, True)]
    for name, T0, steps, warm in cases:
        ag, _ = make_agent(ns, "score", sampler="ode", sampling_steps=steps)
        times, restore = score_time_log(ag)
        init_x = None
        if warm:
            gi = torch.Generator().manual_seed(5)
            r6 = go.normalize_rotation(torch.randn(2, 6, generator=gi))
            init_x = torch.cat([r6, 0.02 * torch.randn(2, 3, generator=gi)], dim=-1)
            g6[f"{name}_init_x"] = init_x.numpy()
        torch.manual_seed(100)
        with DrawLog() as dl:
            data = {"pts": pts2.clone(), "pts_center": cen2.clone()}
            pred, proc = ag.pred_func(data, repeat_num=10, save_path=None, T0=T0, init_x=init_x, return_process=True)
        restore()
        assert len(dl.draws) == 1
        g6[f"{name}_prior_noise"] = dl.draws[0].numpy()  # standard normal [20,9] (before * sigma(T0))
        g6[f"{name}_pred"] = pred.numpy()
        g6[f"{name}_proc_shape"] = np.array(proc.shape)
        g6[f"{name}_proc_last3"] = proc[:, :, -3:].numpy()
        g6[f"{name}_proc_first2"] = proc[:, :, :2].numpy()
        g6[f"{name}_eval_t"] = np.array(times)
        g6[f"{name}_T0"] = np.array(T0)
        g6[f"{name}_steps"] = np.array(-1 if steps is None else steps)
    np.savez_compressed(os.path.join(OUT, "g6_ode.npz"), **g6)

    # ---------------- G7 PC sampler, 20 steps
    ag, _ = make_agent(ns, "score", sampler="pc", sampling_steps=20)
    torch.manual_seed(200)
    with DrawLog() as dl:
        data = {"pts": pts2.clone(), "pts_center": cen2.clone()}
        pred, proc = ag.pred_func(data, repeat_num=10, save_path=None, return_process=True)
    assert len(dl.draws) == 41
    g7 = {"pts": pts2.numpy(), "prior_noise": dl.draws[0].numpy(),
          "z_langevin": torch.stack(dl.draws[1::2]).numpy(), "z_predictor": torch.stack(dl.draws[2::2]).numpy(),
          "pred": pred.numpy(), "proc": proc.numpy()}
    np.savez_compressed(os.path.join(OUT, "g7_pc.npz"), **g7)

    # ---------------- G8 energy + ranking + aggregation (on the T055_none ODE poses)
    pred = torch.from_numpy(g6["T055_none_pred"])
    data = {"pts": pts2.clone(), "pts_center": cen2.clone()}
    energy = e_agent.get_energy(data=data, pose_samples=pred, T=1e-5)
    sorted_pose, sorted_energy = ns.sort_poses_by_energy(pred, energy)
    RT_sorted = go.pose9_to_RT(sorted_pose)  # evaluation_single.py:345-352 (get_rot_matrix is the shimmed pytorch3d)
    RT_unsorted = go.pose9_to_RT(pred)
    # evaluation path (sgpa_utils.py:897-954) re-sorts the *unsorted* hypotheses by energy on the host;
    # its .cuda() call is redirected to CPU for this run only.
    torch.Tensor.cuda = lambda self, *a, **k: self
    sel, avg_sRT, sel_e = ns.sgpa.sort_sRT_by_energy(RT_unsorted.copy(), energy.numpy(), ranker="energy_ranker", ratio=0.6,
                                                     error_mode="average")
    del torch.Tensor.cuda
    g8 = {"pts": pts2.numpy(), "pred": pred.numpy(), "energy": energy.numpy(), "sorted_pose": sorted_pose.numpy(),
          "sorted_energy": sorted_energy.numpy(), "RT_sorted": RT_sorted, "selected_sRT": sel, "average_sRT": avg_sRT,
          "selected_energy": sel_e}
    np.savez_compressed(os.path.join(OUT, "g8_rank.npz"), **g8)

    # ---------------- G9 tracking: 3 frames, 2 objects, warm start from previous average (evaluation_tracking.py:262-337)
    ag, _ = make_agent(ns, "score", sampler="ode", sampling_steps=None)
    frames = synth.golden_tracking_frames()  # [3,2,1024,3]
    gt = synth.golden_tracking_gt()  # [2,4,4] float32
    g9 = {"frames": frames, "gt_RT": gt}
    torch.manual_seed(300)
    prev = None
    torch.Tensor.cuda = lambda self, *a, **k: self
    for fi in range(frames.shape[0]):
        pts = torch.from_numpy(frames[fi])
        cen = pts.mean(dim=1)
        with DrawLog() as dl:
            init_RT = ns.tracking.add_noise_to_RT(torch.from_numpy(gt))  # drawn every frame (:302), then overwritten
            if prev is not None:
                init_RT = prev.clone()
            init_x = torch.cat([init_RT[:, :3, 0], init_RT[:, :3, 1], init_RT[:, :3, 3] - cen], dim=-1).float()
            data = {"pts": pts.clone(), "pts_center": cen.clone()}
            pred = ag.pred_func(data, repeat_num=10, save_path=None, T0=0.15, init_x=init_x)
        energy = e_agent.get_energy(data={"pts": pts.clone(), "pts_center": cen.clone()}, pose_samples=pred, T=1e-5)
        sp, se = ns.sort_poses_by_energy(pred, energy)
        RTs = go.pose9_to_RT(sp)
        avg = _cal_average(ns, RTs, max(1, int(0.6 * 10)))
        g9[f"f{fi}_init_x"] = init_x.numpy()
        g9[f"f{fi}_prior_noise"] = dl.draws[-1].numpy()
        for di in range(4):  # add_noise_to_RT draws: theta [B], quaternion [B,4], norm [B], direction [B,3]
            g9[f"f{fi}_noise_draw{di}"] = dl.draws[di].numpy()
        g9[f"f{fi}_n_draws"] = np.array(len(dl.draws))
        g9[f"f{fi}_pred"] = pred.numpy()
        g9[f"f{fi}_energy"] = energy.numpy()
        g9[f"f{fi}_avg_sRT"] = avg.numpy()
        prev = avg
    del torch.Tensor.cuda
    np.savez_compressed(os.path.join(OUT, "g9_track.npz"), **g9)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def _cal_average(ns, RTs, sel):
    """cal_average_sRT (evaluation_tracking.py:60-77) lives in a runner module that cannot be imported
    (module-level get_config/makedirs/.cuda side effects); its body is the same sequence of
    reference/pytorch3d calls as sort_sRT_by_energy's 'average' tail, which IS importable - use that."""
    e = np.zeros(RTs.shape[:2] + (2,))
    e[:, :, :] = -np.arange(RTs.shape[1])[None, :, None]  # already sorted: keep order
    _, avg, _ = ns.sgpa.sort_sRT_by_energy(RTs.copy(), e, ranker="energy_ranker", ratio=sel / RTs.shape[1] + 1e-9,
                                           error_mode="average")
    return torch.from_numpy(avg).float()


def main_g10():
    """G10 mAP evaluation: the reference's compute_mAP (utils/sgpa_utils.py:957-1183) with the threshold grids of
    evaluate() (runners/evaluation_single.py:493-495, 538-542) on synthetic detection / multi-hypothesis results."""
    import copy
    import tempfile
    from genpose_amd import synth
    ns = ref_import.load()
    K = 10
    results = synth.golden_map_results(77, 12, K)
    degree = list(range(0, 46, 1))
    shift = [i / 2 for i in range(21)]
    iou = [i / 100 for i in range(101)]
    g10 = {"K": np.array(K)}
    torch.Tensor.cuda = lambda self, *a, **k: self  # sort_sRT_by_energy moves its quaternions to the GPU (sgpa_utils.py:938)
    try:
        for mode in ("average", "nearest"):
            with tempfile.TemporaryDirectory() as d:
                iou_aps, pose_aps, iou_acc, pose_acc = ns.sgpa.compute_mAP(
                    copy.deepcopy(results), d, degree, shift, iou, iou_pose_thres=0.1, use_matches_for_pose=True, repeat_num=K,
                    pooling_mode=mode, ratio=0.6, so3_vis=False, ranker="energy_ranker")
            g10[f"{mode}_iou_aps"], g10[f"{mode}_pose_aps"] = iou_aps, pose_aps
            g10[f"{mode}_iou_acc"], g10[f"{mode}_pose_acc"] = iou_acc, pose_acc
        # building blocks on the first image with >= 2 ground-truth objects and >= 2 detections
        r = next(x for x in results if len(x["gt_class_ids"]) >= 2 and len(x["pred_class_ids"]) >= 2)
        gm, pm, ov, idx = ns.sgpa.compute_2d_IoU_matches(r["gt_class_ids"], r["gt_bboxes"], r["pred_class_ids"], r["pred_bboxes"],
                                                        r["pred_scores"], iou)
        g10["blk_gt_matches"], g10["blk_pred_matches"], g10["blk_overlaps"], g10["blk_indices"] = gm, pm, ov, idx
        sel, avg, sel_e = ns.sgpa.sort_sRT_by_energy(r["multi_hypothesis_pred_RTs"].copy(), r["energy"].copy(), None, "energy_ranker", 0.6, "average")
        g10["blk_selected"], g10["blk_average"], g10["blk_selected_energy"] = sel, avg, sel_e
        syn = ["BG", "bottle", "bowl", "camera", "can", "laptop", "mug"]
        rt = ns.sgpa.compute_RT_overlaps(r["gt_class_ids"], r["gt_RTs"], r["gt_handle_visibility"], r["pred_class_ids"], avg, syn)
        g10["blk_RT_overlaps"] = rt
        pgm, ppm = ns.sgpa.compute_RT_matches(rt, r["pred_class_ids"], r["gt_class_ids"], degree + [360], shift + [100])
        g10["blk_pose_gt_matches"], g10["blk_pose_pred_matches"] = pgm, ppm
    finally:
        del torch.Tensor.cuda
    np.savez_compressed(os.path.join(OUT, "g10_map.npz"), **g10)
    print("g10_map.npz", os.path.getsize(os.path.join(OUT, "g10_map.npz")))


def main_g11():
    """G11 depth + mask -> cloud: the importable reference pieces (get_bbox, get_2d_coord_np, crop_resize_by_warp_affine /
    get_affine_transform on the shimmed cv2) composed exactly as detect_mrcnn_genpose does (evaluation_single.py:165-216);
    depth_to_pcl / sample_points are nested functions there and come from oracle/preprocess_oracle.py."""
    from genpose_amd import synth
    from oracle import preprocess_oracle as po
    ns = ref_import.load()
    sys.path.insert(0, ref_import.REF)
    from utils import datasets_utils as du
    import cv2  # the shim
    depth, masks, rois, class_ids = synth.golden_depth_frame()
    K = po.REAL_INTRINSICS
    g = {"intrinsics": K}
    im_H, im_W = depth.shape
    for i in range(len(class_ids)):
        rmin, rmax, cmin, cmax = ns.sgpa.get_bbox(rois[i])
        assert (rmin, rmax, cmin, cmax) == po.get_bbox(rois[i])
        g[f"i{i}_bbox"] = np.array([rmin, rmax, cmin, cmax])
        mask = np.logical_and(masks[:, :, i], depth > 0)
        coord_2d = du.get_2d_coord_np(im_W, im_H).transpose(1, 2, 0)
        x1, y1, x2, y2 = cmin, rmin, cmax, rmax
        center = np.array([0.5 * (x1 + x2), 0.5 * (y1 + y2)])
        scale = min(max(y2 - y1, x2 - x1), max(im_H, im_W)) * 1.0
        roi_coord = du.crop_resize_by_warp_affine(coord_2d, center, scale, 256, interpolation=cv2.INTER_NEAREST).transpose(2, 0, 1)
        roi_mask = du.crop_resize_by_warp_affine(mask.copy().astype(np.float32), center, scale, 256, interpolation=cv2.INTER_NEAREST)
        roi_depth = du.crop_resize_by_warp_affine(depth, center, scale, 256, interpolation=cv2.INTER_NEAREST)
        g[f"i{i}_roi_depth_sum"] = np.array(int(roi_depth.astype(np.int64).sum()))
        g[f"i{i}_roi_mask_sum"] = np.array(int(roi_mask.sum()))
        g[f"i{i}_roi_coord_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(roi_coord).tobytes()).digest(), dtype=np.uint8)
        valid = ((roi_depth.reshape(-1).astype(np.float32) > 0) * roi_mask.reshape(-1)) > 0
        g[f"i{i}_n_valid"] = np.array(int(valid.sum()))
        cloud = po.instance_cloud(depth, masks[:, :, i], rois[i], K)
        if cloud is None:
            assert np.sum(roi_depth > 0) <= 1 or valid.sum() <= 1
            continue
        # the oracle's crop equals the reference functions' crop
        assert cloud.shape[0] == int(valid.sum())
        d = roi_depth.reshape(-1).astype(np.float32)[valid]
        assert np.array_equal(cloud[:, 2], d / np.float32(1000.0))
        assert np.array_equal(cloud[:, 0], ((roi_coord[0].reshape(-1)[valid] - K[0, 2]) * d / K[0, 0]) / np.float32(1000.0))
        g[f"i{i}_cloud_head"] = cloud[:64]
        g[f"i{i}_cloud_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(cloud).tobytes()).digest(), dtype=np.uint8)
    rs = np.random.RandomState(11)
    pts, cat, inst = po.frame_clouds(depth, masks, rois, class_ids, K, rng=rs)
    g["points"], g["cat_id"], g["valid_inst"] = pts, np.array(cat), np.array(inst)
    np.savez_compressed(os.path.join(OUT, "g11_preprocess.npz"), **g)
    print("g11_preprocess.npz", os.path.getsize(os.path.join(OUT, "g11_preprocess.npz")))


def main_g12():
    """G12 likelihood: GFObjectPose.forward(mode='likelihood') of the imported reference (posenet.py:133-147, samplers.py:22-99)
    on 3 clouds, the Hutchinson probe pinned by replacing prior_fn for the call."""
    from genpose_amd import synth
    ns = ref_import.load()
    ag, sd = make_agent(ns, "score")
    pts = torch.from_numpy(synth.make_batch(3, start=70))
    data = {"pts": pts.clone(), "pts_center": pts.mean(dim=1)}
    with torch.no_grad():
        data["pts_feat"] = ag.net(data, mode="pts_feature")
    gen = torch.Generator().manual_seed(12)
    pose = torch.randn(3, 9, generator=gen)
    pose[:, :6] = go.normalize_rotation(pose[:, :6])
    pose[:, 6:] = pts.mean(dim=1) + 0.02 * torch.randn(3, 3, generator=gen)
    probe = torch.randn(3, 9, generator=gen) * 50.0
    saved = ag.net.prior_fn
    ag.net.prior_fn = lambda shape, **k: probe.clone()
    data["sampled_pose"] = pose.clone()
    ll = ag.net(data, mode="likelihood")
    ag.net.prior_fn = saved
    z, ll_o, nfev = go.ode_likelihood(sd, data["pts_feat"], pose, probe)
    assert np.allclose(ll.numpy(), ll_o.numpy(), rtol=1e-9, atol=1e-9), (ll, ll_o)  # oracle == reference
    g12 = {"pts": pts.numpy(), "pose": pose.numpy(), "probe": probe.numpy(), "log_likelihood": ll.numpy(), "z": z.numpy(), "nfev": np.array(nfev)}
    np.savez_compressed(os.path.join(OUT, "g12_likelihood.npz"), **g12)
    print("g12_likelihood.npz", ll.numpy(), "nfev", nfev)


def main_g13():
    """G13 score of the energy model: PoseEnergyNet.forward(return_item='score') of the imported reference (energynet.py:200-222)
    through GFObjectPose.forward(mode='score') with posenet_mode='energy' (posenet.py:154-157)."""
    ns = ref_import.load()
    ag, sd = make_agent(ns, "energy")
    gen = torch.Generator().manual_seed(13)
    pf = torch.randn(6, 1024, generator=gen).abs()
    pose = torch.randn(6, 9, generator=gen)
    g13 = {"pts_feat": pf.numpy(), "pose": pose.numpy(), "t": np.array([1e-5, 0.15, 0.7])}
    for i, t in enumerate(g13["t"]):
        data = {"pts_feat": pf.clone(), "sampled_pose": pose.clone(), "t": torch.ones(6, 1) * float(t)}
        sc = ag.net(data, mode="score").detach()
        ref_s, ref_e = go.energy_score(sd, pf, pose, torch.ones(6, 1) * float(t))
        assert np.allclose(sc.numpy(), ref_s.numpy(), rtol=1e-5, atol=1e-5 * float(ref_s.abs().max()))
        g13[f"score_{i}"] = sc.numpy()
    np.savez_compressed(os.path.join(OUT, "g13_energy_score.npz"), **g13)
    print("g13_energy_score.npz", os.path.getsize(os.path.join(OUT, "g13_energy_score.npz")))


def main_g14():
    """G14 one training step of the score model: the imported reference's own train_score_func pieces (posenet_agent.py:285-305:
    net.train(), pts_feature, collect_score_loss -> losses.py:47-89, update_network -> Adam + grad clipping, ema.update) on 4 clouds,
    repeat_num = 2, with the loss's torch.rand / torch.randn_like draws logged.  (train_score_func itself also writes to a
    tensorboard writer that only exists with cfg.is_train.)"""
    ns = ref_import.load()
    from networks.gf_algorithms.score_utils import ExponentialMovingAverage  # the reference's own (importable after load())
    ag, sd = make_agent(ns, "score")
    ag.cfg.repeat_num, ag.cfg.grad_clip = 2, 1.0
    ag.ema = ExponentialMovingAverage(ag.net.parameters(), decay=ag.cfg.ema_rate)  # built on the loaded weights
    B = 4
    pts = torch.from_numpy(synth.make_batch(B, start=800))
    centre = pts.mean(dim=1)
    gen = torch.Generator().manual_seed(14)
    a = torch.nn.functional.normalize(torch.randn(B, 3, generator=gen), dim=-1)
    b = torch.randn(B, 3, generator=gen)
    b = torch.nn.functional.normalize(b - (a * b).sum(-1, keepdim=True) * a, dim=-1)
    gt = torch.cat([a, b, 0.05 * torch.randn(B, 3, generator=gen)], dim=-1)
    data = {"pts": pts, "zero_mean_pts": pts - centre.unsqueeze(1), "pts_center": centre, "zero_mean_gt_pose": gt}
    draws_u, draws_z = [], []
    _rand, _randn_like = torch.rand, torch.randn_like

    def rand(*a_, **k):
        r = _rand(*a_, **k)
        draws_u.append(r.detach().clone())
        return r

    def randn_like(x, **k):
        r = _randn_like(x, **k)
        draws_z.append(r.detach().clone())
        return r

    torch.manual_seed(1414)
    torch.set_grad_enabled(True)
    torch.rand, torch.randn_like = rand, randn_like
    try:
        ag.net.train()
        data["pts_feat"] = ag.net(data, mode="pts_feature")
        losses = ag.collect_score_loss(data)
        ag.update_network(losses)
        ag.ema.update(ag.net.parameters())
    finally:
        torch.rand, torch.randn_like = _rand, _randn_like
        torch.set_grad_enabled(False)
    assert len(draws_u) == 2 and len(draws_z) == 2
    names = [n for n, p_ in ag.net.named_parameters() if p_.requires_grad]
    params = dict(ag.net.named_parameters())
    shadow = dict(zip(names, ag.ema.shadow_params))
    g14 = {"pts": pts.numpy(), "gt_pose": gt.numpy(), "u": torch.stack(draws_u).numpy(), "z": torch.stack(draws_z).numpy(),
           "loss": np.float64(losses["gf"].item()), "param_names": np.array(names),
           "grad_norms": np.array([float(params[n].grad.norm()) for n in names]),
           "lr": np.float64(ag.cfg.lr), "ema_rate": np.float64(ag.cfg.ema_rate)}
    for tag, n in (("enc0", "pts_encoder.SA_modules.0.mlps.0.layer0.conv.weight"), ("enc3bn", "pts_encoder.SA_modules.3.mlps.1.layer2.bn.bn.weight"),
                   ("pose0", "pose_score_net.pose_encoder.0.weight"), ("tail", "pose_score_net.fusion_tail_trans.2.weight")):
        g14[f"{tag}_grad"] = params[n].grad.numpy().copy()
        g14[f"{tag}_new"] = params[n].detach().numpy().copy()
        g14[f"{tag}_ema"] = shadow[n].numpy().copy()
    bn = ag.net.pts_encoder.SA_modules[0].mlps[0].layer0.bn.bn
    g14["bn0_running_mean"], g14["bn0_running_var"] = bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g14_train_step.npz"), **g14)
    print("g14_train_step.npz", os.path.getsize(os.path.join(OUT, "g14_train_step.npz")), "loss", g14["loss"])


def main_g15():
    """G15 one training step of the ENERGY model with the ranking loss: the imported reference's own train_energy_func pieces
    (posenet_agent.py:262-283: net.train(), pts_feature, collect_score_loss on PoseEnergyNet's autograd score, collect_ranking_loss
    -> get_energy(mode='train') / get_metrics / sort_results / ranking_loss, update_network, ema.update) on 4 clouds x 6 candidate
    poses, repeat_num = 2, with the torch.rand / torch.randn_like / torch.randint draws logged."""
    ns = ref_import.load()
    from networks.gf_algorithms.score_utils import ExponentialMovingAverage
    ag, sd = make_agent(ns, "energy")
    ag.cfg.repeat_num, ag.cfg.grad_clip = 2, 1.0
    ag.ema = ExponentialMovingAverage(ag.net.parameters(), decay=ag.cfg.ema_rate)
    B, K = 4, 6
    pts = torch.from_numpy(synth.make_batch(B, start=1500))
    centre = pts.mean(dim=1)
    gen = torch.Generator().manual_seed(15)

    def rand_pose(n, trans_scale):
        a = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
        b = torch.randn(n, 3, generator=gen)
        b = torch.nn.functional.normalize(b - (a * b).sum(-1, keepdim=True) * a, dim=-1)
        return torch.cat([a, b, trans_scale * torch.randn(n, 3, generator=gen)], dim=-1)

    gt0 = rand_pose(B, 0.05)                        # zero-mean translation
    gt = gt0.clone()
    gt[:, 6:] += centre                              # camera-frame ground truth
    noise = 0.25 * torch.randn(B, K, 9, generator=gen)
    noise[:, :, 6:] *= 0.1
    cand = gt.unsqueeze(1) + noise                   # candidates of a (pretend) score model: perturbed ground truth
    r6 = ns.misc.normalize_rotation(cand.reshape(B * K, 9)[:, :6].clone(), "rot_matrix") if hasattr(ns.misc, "normalize_rotation") else None
    cand = cand.reshape(B * K, 9)
    if r6 is not None:
        cand[:, :6] = r6
    cand = cand.reshape(B, K, 9)
    ids = torch.tensor([0, 2, 5, 5])                 # bottle (symmetric), camera, mug without / with a visible handle
    vis = torch.tensor([1, 1, 0, 1])
    data = {"pts": pts, "zero_mean_pts": pts - centre.unsqueeze(1), "pts_center": centre, "zero_mean_gt_pose": gt0, "gt_pose": gt,
            "id": ids, "handle_visibility": vis}
    draws_u, draws_z, draws_t = [], [], []
    _rand, _randn_like, _randint = torch.rand, torch.randn_like, torch.randint

    def rand(*a_, **k):
        r = _rand(*a_, **k)
        draws_u.append(r.detach().clone())
        return r

    def randn_like(x, **k):
        r = _randn_like(x, **k)
        draws_z.append(r.detach().clone())
        return r

    def randint(*a_, **k):
        r = _randint(*a_, **k)
        draws_t.append(r.detach().clone())
        return r

    torch.manual_seed(1515)
    torch.set_grad_enabled(True)
    torch.rand, torch.randn_like, torch.randint = rand, randn_like, randint
    try:
        ag.net.train()
        ag.is_testing = False
        data["pts_feat"] = ag.net(data, mode="pts_feature")
        ag.pts_feature = True
        losses = {**ag.collect_score_loss(data), **ag.collect_ranking_loss(data, cand)}
        ag.update_network(losses)
        ag.ema.update(ag.net.parameters())
    finally:
        torch.rand, torch.randn_like, torch.randint = _rand, _randn_like, _randint
        torch.set_grad_enabled(False)
    assert len(draws_u) == 2 and len(draws_z) == 2 and len(draws_t) == 1
    from utils.metrics import get_metrics
    rot_err, trans_err = get_metrics(cand.reshape(B * K, 9), gt.unsqueeze(1).repeat(1, K, 1).reshape(B * K, 9),
                                     class_ids=ids.unsqueeze(1).repeat(1, K).reshape(-1, 1), synset_names=ag.cfg.synset_names,
                                     gt_handle_visibility=vis.unsqueeze(1).repeat(1, K).reshape(-1, 1), pose_mode="rot_matrix", o2c_pose=ag.cfg.o2c_pose)
    names = [n for n, p_ in ag.net.named_parameters() if p_.requires_grad]
    params = dict(ag.net.named_parameters())
    shadow = dict(zip(names, ag.ema.shadow_params))
    g15 = {"pts": pts.numpy(), "gt_pose": gt.numpy(), "zero_mean_gt_pose": gt0.numpy(), "pose_samples": cand.numpy(), "id": ids.numpy(),
           "handle_visibility": vis.numpy(), "u": torch.stack(draws_u).numpy(), "z": torch.stack(draws_z).numpy(), "t_draws": draws_t[0].numpy(),
           "loss_gf": np.float64(losses["gf"].item()), "loss_ranking": np.float64(losses["ranking"].item()),
           "rot_err": np.asarray(rot_err, dtype=np.float64), "trans_err": np.asarray(trans_err, dtype=np.float64),
           "param_names": np.array(names), "grad_norms": np.array([float(params[n].grad.norm()) for n in names]),
           "lr": np.float64(ag.cfg.lr), "ema_rate": np.float64(ag.cfg.ema_rate)}
    for tag, n in (("enc0", "pts_encoder.SA_modules.0.mlps.0.layer0.conv.weight"), ("pose0", "pose_score_net.pose_encoder.0.weight"),
                   ("tail", "pose_score_net.fusion_tail_trans.2.weight"), ("head", "pose_score_net.fusion_tail_rot_x.0.weight")):
        cut = slice(0, 8) if tag == "head" else slice(None)  # [256,1408]: the first rows do
        g15[f"{tag}_grad"] = params[n].grad.numpy()[cut].copy()
        g15[f"{tag}_new"] = params[n].detach().numpy()[cut].copy()
        g15[f"{tag}_ema"] = shadow[n].numpy()[cut].copy()
    np.savez_compressed(os.path.join(OUT, "g15_energy_train_step.npz"), **g15)
    print("g15_energy_train_step.npz", os.path.getsize(os.path.join(OUT, "g15_energy_train_step.npz")), "gf", g15["loss_gf"], "ranking", g15["loss_ranking"])


def main_g16(params):
    """G16: the encoder under the reference's OTHER configurations (--pointnet2_params dense | lighter, pointnet2.py:47-78; the reference
    picks the configuration when networks/pts_encoder/pointnet2.py is imported, so every configuration is its own process):
    Pointnet2ClsMSG(0) forward on the four golden clouds, per-level features of cloud 0."""
    from genpose_amd.weights import ENCODER_CFGS
    from genpose_amd.weights_synth import make_state_dict
    os.makedirs(OUT, exist_ok=True)
    ns = ref_import.load(extra_argv=("--pointnet2_params", params))
    torch.set_grad_enabled(False)
    import argparse
    cfg = argparse.Namespace(**vars(ns.cfg))
    assert cfg.pointnet2_params == params
    cfg.posenet_mode, cfg.sampler_mode, cfg.sampling_steps = "score", ["ode"], None
    agent = ns.PoseNet(cfg)
    agent.net.load_state_dict(make_state_dict(0, "score", params), strict=True)
    agent.net.eval()
    clouds = synth.golden_clouds()
    xyz = torch.from_numpy(clouds)
    feat = agent.net({"pts": xyz}, mode="pts_feature")
    inter = []
    hooks = [m.register_forward_hook(lambda m, i, o: inter.append(o)) for m in agent.net.pts_encoder.SA_modules]
    agent.net({"pts": xyz[:1]}, mode="pts_feature")
    for h in hooks:
        h.remove()
    g = {"clouds": clouds, "feat": feat.numpy(), "seed": np.array(0), "params": np.array(params)}
    nlev = sum(1 for n in ENCODER_CFGS[params]["npoints"] if n is not None)
    assert len(inter) == nlev + 1
    for lvl in range(nlev):
        nx, f = inter[lvl]
        g[f"new_xyz{lvl}"] = nx[0].numpy()
        g[f"feat{lvl}_first32"] = f[0, :, :32].numpy()  # [C, 32 points]
    np.savez_compressed(os.path.join(OUT, f"g16_encoder_{params}.npz"), **g)
    print("wrote", f"g16_encoder_{params}.npz", feat.shape, float(feat.abs().mean()))


def main_arith(arith):
    """G1/G2 only, under a non-default convention (python -m oracle.gen_golden --arith A|B|C)."""
    ns = ref_import.load()
    torch.set_grad_enabled(False)
    gen_g1_g2(ns, arith)


if __name__ == "__main__":
    if "--arith" in sys.argv:
        sys.exit(main_arith(sys.argv[sys.argv.index("--arith") + 1]))
    if "--g16" in sys.argv:
        sys.exit(main_g16(sys.argv[sys.argv.index("--g16") + 1]))
    if "--g15" in sys.argv:
        sys.exit(main_g15())
    if "--g14" in sys.argv:
        sys.exit(main_g14())
    if "--g13" in sys.argv:
        sys.exit(main_g13())
    if "--g12" in sys.argv:
        sys.exit(main_g12())
    sys.exit(main_g11() if "--g11" in sys.argv else (main_g10() if "--g10" in sys.argv else main()))
