"""GPU parity OUTSIDE the seed-0 random-weight regime: the score model and the energy model trained on synthetic posed clouds
(scratch/train_synth.py: the reference-pinned training step of genpose_amd/training.py looped on the device for a bounded budget; the
checkpoints under tests/golden/trained/ are reference-layout files, what PoseNet.load_ckpt reads) - trained BatchNorm statistics,
trained output layers, a sampler that pulls candidates into modes instead of walking at random.

  checkpoints      reference key schema, loaded through PoseNet.load_ckpt AND into the oracle (the same state dict)
  configs[1] batch 64 held-out clouds x 50 candidates x 100 PC steps against the oracle - the tolerance of the random-weight test
  eval_single      256 held-out clouds, ODE from T0 = 0.55 over 12 800 coupled rows: evaluation count, accept / reject schedule, poses of every
                   row, energies, exact ranking permutation, aggregation - the tolerances of the random-weight tests
  accuracy proxy   5deg2cm / 5deg5cm / 10deg2cm / 10deg5cm of evaluation.compute_mAP on held-out synthetic instances with known poses: HIP path
                   against the oracle with identical draws (north star: within 0.5 pt), HIP path with its own draws, opt-in split-bf16 forms
Every tolerance is imported from the random-weight tests: if one had to move for trained weights, that would be the finding.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go
from oracle import parallel as opar

from test_gpu_fullsize import ENC_ATOL, ENC_RTOL, _assert_pc100_close, _check_pose_properties, _host_threads
from test_gpu_sampler import ODE_ROT_ATOL, ODE_RTOL

HERE = os.path.dirname(os.path.abspath(__file__))
CKPT = {m: os.path.join(HERE, "golden", "trained", f"ckpt_{m}.pth") for m in ("score", "energy")}
HELD_OUT = 1_000_000  # synth.make_posed_cloud indices the training run never saw (it used 0 .. 49 151)
K, T0, RATIO = 50, 0.55, 0.6
REPORT = os.environ.get("GP_PROXY_REPORT")  # path: write the accuracy-proxy table there (profiles/r6_accuracy_proxy.txt)
N_PROXY = int(os.environ.get("GP_PROXY_INSTANCES", "256"))


def _sd(mode):
    assert os.path.exists(CKPT[mode]), f"{CKPT[mode]} is missing: the trained checkpoints are committed fixtures (scratch/train_synth.py makes them)"
    return {k: v.float() for k, v in torch.load(CKPT[mode], map_location="cpu")["model_state_dict"].items()}


def _agent(mode, sampler="ode", steps=None, **cfg):
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    a = PoseNet(get_config(posenet_mode=mode, sampler_mode=[sampler], sampling_steps=steps, **cfg))
    a.load_ckpt(model_dir=CKPT[mode], model_path=True, load_model_only=True)
    return a


def _posed(n, start=HELD_OUT):
    from genpose_amd import synth
    return synth.posed_batch(range(start, start + n))


def test_checkpoints_are_reference_layout_and_trained():
    sd, sde = _sd("score"), _sd("energy")
    ref_keys = set(go.make_state_dict(0, "score"))
    assert set(sd) == ref_keys and set(sde) == ref_keys
    for d in (sd, sde):
        # zero-initialised output layers (scorenet.py:156-170) have moved, BatchNorm statistics are no longer (0, 1), nothing is NaN
        assert all(torch.isfinite(v.float()).all() for v in d.values())
        assert float(d["pose_score_net.fusion_tail_trans.2.weight"].abs().max()) > 1e-3
        rv = d["pts_encoder.SA_modules.2.mlps.0.layer1.bn.bn.running_var"]
        assert float((rv - 1).abs().max()) > 0.05 and int(d["pts_encoder.SA_modules.0.mlps.0.layer0.bn.bn.num_batches_tracked"]) > 100
    a = _agent("score")
    w = a.net.pose_score_net  # the agent holds the file's weights (spot check through the oracle below; here: it loaded at all)
    assert w is not None


def test_config1_batch_pc100_trained():
    """BASELINE configs[1] geometry (64 clouds x 50 candidates x 100 PC steps, injected draws) on the TRAINED score model: agent against
    the CPU oracle holding the same state dict."""
    B, n = 64, 100
    d = _posed(B, HELD_OUT + 5000)
    pts = torch.from_numpy(d["pts"])
    gen = torch.Generator().manual_seed(31)
    prior = torch.randn(B * K, 9, generator=gen)
    z1, z2 = torch.randn(n, B * K, 9, generator=gen), torch.randn(n, B * K, 9, generator=gen)
    a = _agent("score", "pc", n)
    a.net.prior_fn = lambda shape, T=1.0: prior * float(go.ve_sigma(1.0))
    dev = pts.cuda()
    data = {"pts": dev, "pts_center": dev.mean(dim=1)}
    got = a.pred_func(data, K, save_path=None, noise=(z1.cuda(), z2.cuda()))
    torch.cuda.synchronize()
    _check_pose_properties(got)
    feat = opar.encoder_features("score", pts, ckpt=CKPT["score"])
    np.testing.assert_allclose(data["pts_feat"].cpu().numpy(), feat, rtol=ENC_RTOL, atol=ENC_ATOL * max(1.0, float(np.abs(feat).max())))
    sd = _sd("score")
    feat_r = torch.from_numpy(feat).repeat_interleave(K, 0)
    cen_r = pts.mean(dim=1).repeat_interleave(K, 0)
    with _host_threads():
        _, ref = go.pc_sampler(lambda x, t: go.score_forward(sd, feat_r, x, t), prior * float(go.ve_sigma(1.0)), cen_r, n, z1, z2)
    _assert_pc100_close(got.cpu().numpy(), ref.reshape(B, K, 9).numpy(), "trained weights, configs[1] batch (3200 rows x 100 steps) vs oracle")


# ---------------------------------------------------------------------------------------------- eval_single shape + accuracy proxy
def _hip_batch(sa, ea, pts, prior):
    """One batch exactly as SingleFrameRunner.infer_tensors runs it (runner.py:52-66), with the prior draw injected (None: the agent's own)."""
    from genpose_amd import reward, rotation
    from genpose_amd.runner import make_batch_sample
    if prior is not None:
        sa.net.prior_fn = lambda shape, T=1.0: prior * float(go.ve_sigma(T))
    sample = make_batch_sample(pts)
    pred = sa.pred_func(data=sample, repeat_num=K, save_path=None, T0=T0)
    energy = ea.get_energy(data=sample, pose_samples=pred, T=1e-5)
    r = reward.rank_aggregate(pred, energy, ratio=RATIO)
    out = {"pred": pred, "energy": energy, "order": r["order"], "sorted_poses": r["sorted_poses"], "sorted_energy": r["sorted_energy"],
           "avg_pose": r["avg_pose"], "feat": sample["pts_feat"], "sorted_RTs": rotation.pose9_to_RT(r["sorted_poses"])}
    out = {k: v.cpu().clone() for k, v in out.items()}
    stats = getattr(sa.net, "last_sampler", None)
    if stats is not None and getattr(stats, "last_stats", None) and "nfev" in stats.last_stats:
        out["nfev"] = int(stats.last_stats["nfev"])
        out["log_acc"] = [bool(x) for x in stats.last_stats["log_acc"]]
    return out


def _oracle_batch(pts_cpu, prior):
    sd, sde = _sd("score"), _sd("energy")
    B = pts_cpu.shape[0]
    feat = torch.from_numpy(opar.encoder_features("score", pts_cpu, ckpt=CKPT["score"]))
    feat_e = torch.from_numpy(opar.encoder_features("energy", pts_cpu, ckpt=CKPT["energy"]))
    cen = pts_cpu.mean(dim=1)
    cen_r = cen.repeat_interleave(K, 0)
    log = []
    with _host_threads():
        feat_r = feat.repeat_interleave(K, 0)
        _, x, nfev = go.ode_sampler(lambda xx, t: go.score_forward(sd, feat_r, xx, t), prior * go.ve_sigma(T0), cen_r, T0, log=log)
        pose = x.clone().float()
        pose[:, -3:] -= cen_r
        energy = go.energy_forward(sde, feat_e.repeat_interleave(K, 0), pose, torch.ones(B * K, 1) * 1e-5).reshape(B, K, 2)
    pred = x.reshape(B, K, 9)
    sorted_poses, sorted_energy = go.sort_poses_by_energy(pred, energy)
    sorted_RT = go.pose9_to_RT(sorted_poses)
    avg_RT, qt = go.aggregate_sorted(sorted_RT, ratio=RATIO)
    return {"pred": pred, "energy": energy, "sorted_poses": sorted_poses, "sorted_energy": sorted_energy, "sorted_RTs": torch.from_numpy(sorted_RT),
            "avg_qt": qt, "feat": feat, "nfev": nfev, "log_acc": [bool(e["accepted"]) for e in log]}


def _map_results(d, sorted_RTs, sorted_energy):
    """One 'image' per instance in the container layout evaluation.compute_mAP consumes (DetectionResults.results()): the detection is
    given (same class, same box), the pose hypotheses ranked by energy as pred_energy_batch stores them."""
    res = []
    box = np.array([[10, 10, 110, 110]], dtype=np.int32)
    for i in range(sorted_RTs.shape[0]):
        gt = np.eye(4)
        gt[:3, :3], gt[:3, 3] = d["R"][i], d["t"][i]
        cls = np.array([int(d["cat"][i]) + 1], dtype=np.int32)
        res.append({"gt_class_ids": cls, "gt_bboxes": box, "gt_RTs": gt[None], "gt_scales": np.ones((1, 3)), "gt_handle_visibility": np.ones(1, dtype=np.int32),
                    "pred_class_ids": cls, "pred_bboxes": box, "pred_scores": np.ones(1), "pred_RTs": np.eye(4)[None], "pred_scales": np.ones((1, 3)),
                    "multi_hypothesis_pred_RTs": np.asarray(sorted_RTs[i], dtype=np.float64)[None], "energy": np.asarray(sorted_energy[i], dtype=np.float64)[None]})
    return res


def _proxy(d, sorted_RTs, sorted_energy):
    from genpose_amd import evaluation
    deg, sh, iou = [5, 10], [2, 5, 10], [0.1]
    iou_aps, pose_aps, _, _ = evaluation.compute_mAP(_map_results(d, sorted_RTs, sorted_energy), None, deg, sh, iou, iou_pose_thres=0.1,
                                                     use_matches_for_pose=True, repeat_num=K, pooling_mode="average", ratio=RATIO, ranker="energy_ranker")
    return evaluation.summary(iou_aps, pose_aps, iou, deg + [360], sh + [100])


@pytest.fixture(scope="module")
def eval_single():
    """N_PROXY held-out instances in batches of 256 (scripts/eval_single.sh): HIP agents and oracle with the SAME prior draws."""
    nb = max(1, N_PROXY // 256)
    d = _posed(256 * nb)
    sa, ea = _agent("score"), _agent("energy")
    gen = torch.Generator().manual_seed(2026)
    hip, ora, priors = [], [], []
    for b in range(nb):
        pts = torch.from_numpy(d["pts"][256 * b:256 * (b + 1)])
        prior = torch.randn(256 * K, 9, generator=gen)
        priors.append(prior)
        hip.append(_hip_batch(sa, ea, pts.cuda(), prior))
        ora.append(_oracle_batch(pts, prior))
    return {"d": d, "hip": hip, "ora": ora, "priors": priors, "agents": (sa, ea)}


def test_eval_single_shape_trained(eval_single):
    """scripts/eval_single.sh's shape (256 clouds, ODE from T0 = 0.55, K = 50, energy ranking, top-60 % aggregation) on trained weights,
    batch 0 against the oracle."""
    h, o = eval_single["hip"][0], eval_single["ora"][0]
    B = 256
    np.testing.assert_allclose(h["feat"].numpy(), o["feat"].numpy(), rtol=ENC_RTOL, atol=ENC_ATOL * max(1.0, float(o["feat"].abs().max())))
    # the adaptive solve: same evaluation count (one attempt of slack), same accept / reject decisions over the common attempts
    assert abs(h["nfev"] - o["nfev"]) <= 6, (h["nfev"], o["nfev"])
    common = min(len(h["log_acc"]), len(o["log_acc"]))
    flips = [i for i in range(common) if h["log_acc"][i] != o["log_acc"][i]]
    print(f"trained ODE T0={T0}: nfev {h['nfev']} (oracle {o['nfev']}), attempts {len(h['log_acc'])} / {len(o['log_acc'])}, rejected {o['log_acc'].count(False)}, flips {flips}")
    assert not flips
    got, ref = h["pred"].numpy().reshape(B * K, 9), o["pred"].numpy().reshape(B * K, 9)
    assert h["pred"].dtype == torch.float64
    _check_pose_properties(h["pred"].float())
    np.testing.assert_allclose(got[:, :6], ref[:, :6], rtol=0, atol=ODE_ROT_ATOL, err_msg="rotation block, 12800 rows")
    np.testing.assert_allclose(got[:, 6:], ref[:, 6:], rtol=0, atol=ODE_RTOL * max(1.0, float(np.abs(ref[:, 6:]).max())), err_msg="translations")
    # energies of the DEVICE's candidates through the oracle's energy network (the poses agree to 2e-3; compare like with like)
    sde = _sd("energy")
    sl = slice(32, 64)
    d = eval_single["d"]
    pts = torch.from_numpy(d["pts"][:256])
    cen = pts.mean(dim=1)
    feat_e = torch.from_numpy(opar.encoder_features("energy", pts, ckpt=CKPT["energy"]))[sl]
    pose = h["pred"][sl].reshape(-1, 9).float().clone()
    pose[:, -3:] -= cen[sl].repeat_interleave(K, 0)
    ref_e = go.energy_forward(sde, feat_e.repeat_interleave(K, 0), pose, torch.ones(pose.shape[0], 1) * 1e-5).reshape(-1, K, 2).numpy()
    np.testing.assert_allclose(h["energy"][sl].numpy(), ref_e, rtol=5e-4, atol=5e-4 * np.abs(ref_e).max())
    # ranking: the exact stable descending permutation of the device's energies on all 256 clouds; sorted tensors consistent with it
    e_cpu, order = h["energy"], h["order"].long()
    for c in range(2):
        srt = torch.sort(e_cpu[:, :, c], dim=1, descending=True, stable=True)
        assert torch.equal(order[:, :, c], srt.indices)
        assert torch.equal(h["sorted_energy"][:, :, c], srt.values)
    # how well separated are trained energies?  (random weights: gaps of 1e-3 relative; a trained energy model spreads them)
    gap = (h["sorted_energy"][:, :-1] - h["sorted_energy"][:, 1:]).abs() / h["sorted_energy"].abs().amax(dim=1, keepdim=True)
    print(f"trained energies: median relative gap between neighbours in the ranking {float(gap.median()):.2e}, smallest {float(gap.min()):.2e}")
    # aggregation of the device's ranking through the oracle
    _, qt = go.aggregate_sorted(go.pose9_to_RT(h["sorted_poses"][sl]), ratio=RATIO)
    a = h["avg_pose"][sl].numpy()
    np.testing.assert_allclose(a[:, 4:], qt.numpy()[:, 4:], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(qt.numpy()[:, 4:]).max())))
    assert np.all(np.abs(np.sum(a[:, :4] * qt.numpy()[:, :4], axis=1)) > 1 - 1e-5)


def test_accuracy_proxy_within_half_a_point(eval_single):
    """north star: '5deg5cm accuracy within +-0.5 pt of reference'.  REAL275 and the shipped checkpoints are unreachable; the proxy is the
    same metric code (evaluation.compute_mAP, pinned to the reference by G10) on held-out synthetic instances with known poses, HIP path
    against the oracle on identical draws.  Also reported (not asserted against the oracle): the HIP path with its own draws, and the two
    opt-in split-bf16 forms."""
    from genpose_amd import rotation
    d, nb = eval_single["d"], len(eval_single["hip"])
    cat = lambda key, runs: np.concatenate([np.asarray(r[key]) for r in runs], 0)
    rows = []
    hip = _proxy(d, cat("sorted_RTs", eval_single["hip"]), cat("sorted_energy", eval_single["hip"]))
    ora = _proxy(d, cat("sorted_RTs", eval_single["ora"]), cat("sorted_energy", eval_single["ora"]))
    rows.append(("HIP path, fp32, draws shared with the oracle", hip))
    rows.append(("CPU oracle, the same draws", ora))
    sa, ea = eval_single["agents"]
    from genpose_amd.sde import init_sde
    sa.net.prior_fn = init_sde("ve")[0]  # back to the agent's own generator
    torch.manual_seed(7)
    own = [_hip_batch(sa, ea, torch.from_numpy(d["pts"][256 * b:256 * (b + 1)]).cuda(), None) for b in range(nb)]
    rows.append(("HIP path, fp32, its own draws", _proxy(d, cat("sorted_RTs", own), cat("sorted_energy", own))))
    # opt-in split-bf16 encoder (both models), ODE sampler, the shared draws
    sb, eb = _agent("score", encoder_precision="bf16x3"), _agent("energy", encoder_precision="bf16x3")
    bf = [_hip_batch(sb, eb, torch.from_numpy(d["pts"][256 * b:256 * (b + 1)]).cuda(), eval_single["priors"][b]) for b in range(nb)]
    rows.append(("HIP path, encoder_precision='bf16x3' (opt-in), shared draws", _proxy(d, cat("sorted_RTs", bf), cat("sorted_energy", bf))))
    # PC sampler, 100 steps (the benched sampler), first 256 instances: fp32, the opt-in split-bf16 PC step and the oracle on the same injected
    # noise.  NOTE what the number says: the reference's predictor step is `mean_x = x + (drift - g^2 grad) * step_size` with step_size =
    # t_0 - t_1 > 0 (samplers.py:146-148) - it moves AGAINST the score - so its PC sampler does not concentrate on the modes of a trained
    # model; the reference evaluates with the ODE sampler (scripts/eval_single.sh).  Reproduced as written: HIP == oracle (asserted to the
    # PC-100 tolerance in test_config1_batch_pc100_trained), and both score about zero here.
    n = 100
    gen = torch.Generator().manual_seed(99)
    from genpose_amd import reward
    from genpose_amd.runner import make_batch_sample
    d0 = {k: v[:256] for k, v in d.items()}
    pts0 = torch.from_numpy(d["pts"][:256])
    prior = torch.randn(256 * K, 9, generator=gen)
    z1, z2 = torch.randn(n, 256 * K, 9, generator=gen), torch.randn(n, 256 * K, 9, generator=gen)
    zdev = (z1.cuda(), z2.cuda())
    pc = {}
    for prec in ("f32", "bf16x3"):
        ag = _agent("score", "pc", n, sampler_precision=prec)
        ag.net.prior_fn = lambda shape, T=1.0: prior * float(go.ve_sigma(1.0))
        sample = make_batch_sample(pts0.cuda())
        pred = ag.pred_func(data=sample, repeat_num=K, save_path=None, noise=zdev)
        energy = ea.get_energy(data=sample, pose_samples=pred, T=1e-5)
        r = reward.rank_aggregate(pred, energy, ratio=RATIO)
        pc[prec] = _proxy(d0, rotation.pose9_to_RT(r["sorted_poses"]).cpu().numpy(), r["sorted_energy"].cpu().numpy())
    sd, sde = _sd("score"), _sd("energy")
    feat_r = eval_single["ora"][0]["feat"].repeat_interleave(K, 0)
    cen_r = pts0.mean(dim=1).repeat_interleave(K, 0)
    with _host_threads():
        _, xo = go.pc_sampler(lambda x, t: go.score_forward(sd, feat_r, x, t), prior * float(go.ve_sigma(1.0)), cen_r, n, z1, z2)
        pose = xo.clone().float()
        pose[:, -3:] -= cen_r
        feat_e = torch.from_numpy(opar.encoder_features("energy", pts0, ckpt=CKPT["energy"])).repeat_interleave(K, 0)
        eo = go.energy_forward(sde, feat_e, pose, torch.ones(256 * K, 1) * 1e-5).reshape(256, K, 2)
    so, seo = go.sort_poses_by_energy(xo.reshape(256, K, 9), eo)
    rows.append(("PC sampler 100 steps, first 256: HIP path fp32", pc["f32"]))
    rows.append(("PC sampler 100 steps, first 256: CPU oracle, the same draws", _proxy(d0, go.pose9_to_RT(so), seo.numpy())))
    rows.append(("PC sampler 100 steps, first 256: HIP sampler_precision='bf16x3' (opt-in)", pc["bf16x3"]))
    # how far apart are the AGGREGATED poses of the two implementations, measured the way the metric measures (sgpa_utils.py:548-560): the full
    # rotation for camera / laptop / mug, the direction of the y axis for the categories that are symmetric about it (their candidates spread
    # around the axis, the quaternion mean's top eigenvector is then near-degenerate in the spin - which the metric ignores)
    ang, shift = [], []
    for b, (h, o) in enumerate(zip(eval_single["hip"], eval_single["ora"])):
        Rh = rotation.quaternion_to_matrix(h["avg_pose"][:, :4].double()).numpy()
        Ro = rotation.quaternion_to_matrix(o["avg_qt"][:, :4].double()).numpy()
        sym = np.isin(d["cat"][256 * b:256 * (b + 1)], (0, 1, 3))
        cos_full = np.clip((np.trace(Rh @ Ro.transpose(0, 2, 1), axis1=1, axis2=2) - 1) / 2, -1, 1)
        cos_y = np.clip(np.sum(Rh[:, :, 1] * Ro[:, :, 1], axis=1), -1, 1)
        ang.append(np.degrees(np.arccos(np.where(sym, cos_y, cos_full))))
        shift.append(np.linalg.norm(h["avg_pose"][:, 4:].numpy().astype(np.float64) - o["avg_qt"][:, 4:].numpy().astype(np.float64), axis=1) * 1000)
    ang, shift = np.concatenate(ang), np.concatenate(shift)
    keys = ["5deg2cm", "5deg5cm", "10deg2cm", "10deg5cm", "10deg10cm"]
    lines = [f"accuracy proxy: {256 * nb} held-out synthetic instances (synth.make_posed_cloud {HELD_OUT}..), K = {K}, ODE T0 = {T0} unless stated, "
             f"top {int(RATIO * 100)} % by energy averaged; evaluation.compute_mAP, mean AP over the six categories, percent",
             f"{'':72s}" + "".join(f"{k:>11s}" for k in keys)]
    for name, s in rows:
        lines.append(f"{name:72s}" + "".join(f"{s[k]:11.2f}" for k in keys))
    delta = {k: hip[k] - ora[k] for k in keys}
    lines.append(f"{'HIP - oracle (ODE, shared draws)':72s}" + "".join(f"{delta[k]:+11.2f}" for k in keys))
    lines.append(f"aggregated pose, HIP vs oracle on shared draws ({len(ang)} instances): rotation apart median {np.median(ang):.2e} deg, p99 {np.quantile(ang, 0.99):.2e}, max {ang.max():.2e} deg "
                 f"(y axis only for the symmetric categories, as the metric measures); translation apart median {np.median(shift):.2e} mm, p99 {np.quantile(shift, 0.99):.2e}, "
                 f"max {shift.max():.2e} mm")
    lines.append("PC rows: the reference's predictor step moves against the score (samplers.py:146-148, sign as written, reproduced) - its PC sampler does "
                 "not concentrate on the modes of a trained model; the reference evaluates with the ODE sampler.")
    print("\n".join(lines))
    if REPORT:
        with open(REPORT, "w") as f:
            f.write("\n".join(lines) + "\n")
    assert abs(delta["5deg5cm"]) <= 0.5, delta
    assert all(abs(v) <= 1.0 for v in delta.values()), delta
    # degrees / millimetres.  A candidate at the top-60 % boundary whose energy differs by 5e-4 between the implementations may be selected on one
    # side only: the aggregate then moves by 1/30 of the distance between two candidates - hence a bound on the 99th percentile and a looser one on the maximum
    assert np.quantile(ang, 0.99) < 0.5 and ang.max() < 3.0 and np.quantile(shift, 0.99) < 1.0 and shift.max() < 5.0, (ang.max(), shift.max())
    assert abs(pc["f32"]["10deg10cm"] - rows[-2][1]["10deg10cm"]) <= 1.0


def test_tracking_sequence_trained():
    """The tracking loop (evaluation_tracking.py:262-337: warm start from the previous frame's aggregated pose, ODE from T0 = 0.15, energy
    ranking, top-60 % aggregation) on TRAINED weights over a synthetic sequence with known poses: every frame of the HIP TrackingRunner (the
    hipGraph path) against the oracle given the same initial poses and prior draws - initial state, all candidates, evaluation count,
    aggregated pose - and the tracker must actually track: its aggregated poses stay near the ground truth (with random weights they wander)."""
    from genpose_amd import rotation, synth
    from genpose_amd.runner import TrackingRunner, add_noise_to_RT
    F, n_obj, Kt, T0t = 6, 5, 50, 0.15
    seq = synth.posed_sequence(3, n_frames=F, n_obj=n_obj)
    sd, sde = _sd("score"), _sd("energy")
    tr = TrackingRunner(_agent("score"), _agent("energy"), repeat_num=Kt, T0=T0t)
    names = [f"obj{o}" for o in range(n_obj)]
    gen = torch.Generator().manual_seed(12)
    gt0 = torch.eye(4).repeat(n_obj, 1, 1)
    gt0[:, :3, :3], gt0[:, :3, 3] = torch.from_numpy(seq["R"][0]).float(), torch.from_numpy(seq["t"][0]).float()
    sym = np.isin(seq["cat"], (0, 1, 3))
    rot_err, tr_err, nfevs = [], [], []
    for f in range(F):
        pts_cpu = torch.from_numpy(seq["pts"][f])
        prior = torch.randn(n_obj * Kt, 9, generator=gen)
        draws = [torch.randn(n_obj, generator=gen), torch.randn(n_obj, 4, generator=gen), torch.randn(n_obj, generator=gen), torch.randn(n_obj, 3, generator=gen)]
        tr.score_agent.net.prior_fn = lambda shape, T=1.0: prior * float(go.ve_sigma(T))
        out = tr.step(pts_cpu.cuda(), names, gt0, noise_draws=draws)
        nfev = int(tr.score_agent.net.last_sampler.last_stats["nfev"])
        nfevs.append(nfev)
        # ---- the oracle on this frame, from the initial poses the runner used (frame 0: the jittered ground truth of the same draws)
        cen = pts_cpu.mean(dim=1)
        init_x = out["init_x"].cpu().float()
        if f == 0:
            jit = add_noise_to_RT(gt0, draws=draws)
            want = torch.cat([jit[:, :3, 0], jit[:, :3, 1], jit[:, :3, 3] - cen], dim=1)
            np.testing.assert_allclose(init_x.numpy(), want.numpy(), atol=1e-6)
        with _host_threads():
            ref, _, ref_nfev = go.pred_func(sd, pts_cpu, cen, Kt, "ode", prior, T0=T0t, init_x=init_x)
            ref_e = go.get_energy(sde, pts_cpu, cen, out["pred_pose"].cpu(), T=1e-5)
        assert abs(nfev - ref_nfev) <= 6, (f, nfev, ref_nfev)
        got = out["pred_pose"].cpu().numpy()
        np.testing.assert_allclose(got[..., :6], ref.numpy()[..., :6], rtol=0, atol=ODE_ROT_ATOL, err_msg=f"frame {f}: rotation block")
        np.testing.assert_allclose(got[..., 6:], ref.numpy()[..., 6:], rtol=0, atol=ODE_RTOL * max(1.0, float(ref[..., 6:].abs().max())), err_msg=f"frame {f}: translations")
        np.testing.assert_allclose(out["energy"].cpu().numpy(), ref_e.numpy(), rtol=5e-4, atol=5e-4 * float(ref_e.abs().max()))
        sp, _ = go.sort_poses_by_energy(out["pred_pose"].cpu(), out["energy"].cpu())
        avg_RT, _ = go.aggregate_sorted(go.pose9_to_RT(sp), ratio=RATIO)
        np.testing.assert_allclose(out["average_sRT"].cpu().numpy(), avg_RT, atol=5e-4, err_msg=f"frame {f}: aggregated pose")
        # ---- against the ground truth of the frame
        Ra = out["average_sRT"][:, :3, :3].cpu().numpy().astype(np.float64)
        Rg = seq["R"][f]
        cos_full = np.clip((np.trace(Ra @ Rg.transpose(0, 2, 1), axis1=1, axis2=2) - 1) / 2, -1, 1)
        cos_y = np.clip(np.sum(Ra[:, :, 1] * Rg[:, :, 1], axis=1), -1, 1)
        rot_err.append(np.degrees(np.arccos(np.where(sym, cos_y, cos_full))))
        tr_err.append(np.linalg.norm(out["average_sRT"][:, :3, 3].cpu().numpy() - seq["t"][f], axis=1) * 100)
    rot_err, tr_err = np.array(rot_err), np.array(tr_err)
    print(f"trained tracker over {F} frames x {n_obj} objects (categories {[synth.CATEGORIES[c] for c in seq['cat']]}): rotation error vs ground truth per frame "
          f"(median over objects) {np.round(np.median(rot_err, axis=1), 1).tolist()} deg, translation {np.round(np.median(tr_err, axis=1), 2).tolist()} cm; "
          f"evaluations per frame {nfevs}")
    assert np.median(tr_err) < 2.0 and np.median(rot_err) < 15.0, (np.median(rot_err), np.median(tr_err))
    assert np.median(tr_err[-1]) < 3.0, "the tracker has drifted away by the last frame"
