"""GPU parity: PoseNet agent surface - ODE sampler (G6), energy + ranking + aggregation (G8), tracking loop (G9)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go

# The PF-ODE is integrated with rtol = atol = 1e-5 PER STEP over 30-60 adaptive steps (samplers.py:168-169), and with
# seeded random weights the flow is not contractive (translations reach |t| ~ 1e2..1e3): two correct implementations
# whose score network differs by fp32 round-off (1e-6 relative, tests/test_gpu_score.py) agree to a few 1e-4 RELATIVE
# of the TRANSLATION scale at the end point.  The rotation block of every returned pose is normalised (two unit columns,
# samplers.py:220-225), so it gets its own ABSOLUTE tolerance - a translation of 2 890 must not hide a rotation error:
#     rotation  (components 0..5): |hip - ref| <= ODE_ROT_ATOL
#     translation (components 6..8): |hip - ref| <= ODE_RTOL * max(1, max|ref translation|)
ODE_RTOL = 5e-4
ODE_ROT_ATOL = 2e-3


def ode_close(got, ref):
    """got / ref [..., 9] poses (rot6 | translation)"""
    np.testing.assert_allclose(got[..., :6], ref[..., :6], rtol=0, atol=ODE_ROT_ATOL, err_msg="rotation block")
    np.testing.assert_allclose(got[..., 6:], ref[..., 6:], rtol=0, atol=ODE_RTOL * max(1.0, float(np.abs(ref[..., 6:]).max())),
                               err_msg="translation block")


def make_agent(mode, sampler="ode", steps=None):
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    agent = PoseNet(get_config(posenet_mode=mode, sampler_mode=[sampler], sampling_steps=steps))
    agent.load_state_dict(go.make_state_dict(0, mode))
    return agent


class FixedPrior:
    """Replaces the CPU-generator draw of ve_prior (sde.py:26-28) by the logged draw of the golden run."""

    def __init__(self, agent, noise):
        self.agent, self.noise = agent, torch.from_numpy(noise)

    def __enter__(self):
        self.saved = self.agent.net.prior_fn
        self.agent.net.prior_fn = lambda shape, T=1.0: self.noise * (0.01 * (50.0 / 0.01) ** T)
        return self

    def __exit__(self, *e):
        self.agent.net.prior_fn = self.saved


@pytest.mark.parametrize("case", ["T1_none", "T055_none", "T055_s20", "T015_warm"])
def test_ode_golden(golden, case):
    g = golden("g6_ode.npz")
    steps = int(g[f"{case}_steps"])
    agent = make_agent("score", "ode", None if steps < 0 else steps)
    pts = torch.from_numpy(g["pts"]).cuda()
    init_x = torch.from_numpy(g[f"{case}_init_x"]).cuda() if f"{case}_init_x" in g else None
    data = {"pts": pts, "pts_center": pts.mean(dim=1)}
    with FixedPrior(agent, g[f"{case}_prior_noise"]):
        want_proc = True  # steps < 0: accepted states; steps given: RK45 dense output at the t_eval points
        out = agent.pred_func(data, repeat_num=10, save_path=None, T0=float(g[f"{case}_T0"]), init_x=init_x, return_process=want_proc)
    pred, proc = out
    assert pred.dtype == torch.float64 and "pts_feat" in data
    ode_close(pred.cpu().numpy(), g[f"{case}_pred"])
    stats = agent.net.last_sampler.last_stats
    ref_nfev = len(g[f"{case}_eval_t"])
    assert stats["status"] == 1
    # The adaptive controller takes the reference's step schedule.  The reference problems are chaotic (random weights, T0 up
    # to 1 with sigma = 50): fp32 rounding of the score (6e-7 relative on either side) is amplified through err^(-1/5) step-size
    # feedback, so the schedules agree closely for the first few dozen attempts and may drift apart late (observed: T0 = 1,
    # error norms agree to 1 % until attempt ~40, first accept/reject flip at attempt 65 of 70).  Checked here: the oracle
    # (CPU restatement, itself pinned to the reference's schedule) and the device agree attempt by attempt over the leading
    # attempts, a flip can only come after them, and the evaluation count stays within 15 %.
    log = []
    prior = torch.from_numpy(g[f"{case}_prior_noise"])
    ix = torch.from_numpy(g[f"{case}_init_x"]) if f"{case}_init_x" in g else None
    go.pred_func(go.make_state_dict(0, "score"), pts.cpu(), pts.cpu().mean(dim=1), 10, "ode", prior, T0=float(g[f"{case}_T0"]),
                 sampling_steps=None if steps < 0 else steps, init_x=ix, log=log)
    acc_ref = _ref_accepts(g[f"{case}_eval_t"])
    assert [bool(e["accepted"]) for e in log][: len(acc_ref)] == acc_ref  # oracle == reference schedule
    acc_dev = [bool(a) for a in stats["log_acc"]]
    first_diff = next((i for i, (a, b) in enumerate(zip(acc_dev, acc_ref)) if a != b), None)
    lead = min(20, len(acc_ref) - 1) if first_diff is None else min(20, first_diff)
    assert first_diff is None or first_diff >= min(20, len(acc_ref) // 2), f"schedules split at attempt {first_diff}"
    for i in range(lead):
        assert abs(stats["log_t"][i] - log[i]["t"]) <= 1e-4 * abs(log[i]["t"]) + 1e-9, i
        assert abs(stats["log_h"][i] - log[i]["h"]) <= 2e-3 * abs(log[i]["h"]), i
        assert abs(stats["log_err"][i] - log[i]["err_norm"]) <= 0.01 * log[i]["err_norm"] + 1e-3, (i, stats["log_err"][i], log[i]["err_norm"])
    assert abs(int(stats["nfev"]) - ref_nfev) <= 0.15 * ref_nfev, (stats["nfev"], ref_nfev)
    if first_diff is None:
        assert abs(int(stats["nfev"]) - ref_nfev) <= 6  # at most the last, ulp-sized step differs
    same_count = int(stats["nfev"]) == ref_nfev
    # ONE attempt more or fewer with an identical accept / reject prefix (observed on the driver's box for the chaotic T0 = 1 problem: 417
    # against 411 evaluations, no flip): the step SIZES drift over the last dozen attempts until one side needs an extra step to reach
    # t_bound.  Everything before the drift is comparable and is compared; where the drift starts is asserted to be late.
    tail_only = (not same_count) and first_diff is None and abs(int(stats["nfev"]) - ref_nfev) <= 6
    common = min(len(acc_dev), len(acc_ref))
    if same_count or tail_only:
        ref_t = g[f"{case}_eval_t"]
        # same accept/reject sequence, and every attempt starts where the reference's did: first stage evaluation of an
        # attempt sits at t + h/5 (Dormand-Prince c_2).  Step sizes follow err^(-1/5), so fp32-level differences in the
        # score move them by ~1e-3 relative late in the integration; the reference logged f32 times.
        assert acc_dev[:common] == acc_ref[:common] and (tail_only or len(acc_dev) == len(acc_ref))
        dev_first_stage = (np.asarray(stats["log_t"]) + 0.2 * np.asarray(stats["log_h"]))[:common]
        ref_first_stage = np.asarray(ref_t[2:-1:6][:common], dtype=np.float64)
        off = np.abs(dev_first_stage - ref_first_stage) > 2e-3 * np.abs(ref_first_stage) + 2e-5
        agree = int(np.argmax(off)) if off.any() else common  # attempts before the first one that starts somewhere else
        assert agree == common if same_count else agree >= 0.75 * common, (agree, common, dev_first_stage[agree:agree + 3], ref_first_stage[agree:agree + 3])
        if proc is not None:
            extra = proc.shape[2] - int(g[f"{case}_proc_shape"][2])  # accepted states the device has beyond the reference's (0 unless tail_only)
            assert [proc.shape[0], proc.shape[1], proc.shape[3]] == [int(v) for v in np.asarray(g[f"{case}_proc_shape"])[[0, 1, 3]]] and abs(extra) <= (1 if tail_only else 0)
            ode_close(proc[:, :, :2].cpu().numpy(), g[f"{case}_proc_first2"])
            if same_count:
                ode_close(proc[:, :, -3:].cpu().numpy(), g[f"{case}_proc_last3"])
    # which branch ran is part of the result: the non-chaotic cases must take the reference's evaluation count exactly (and with it
    # the full-trajectory asserts above); only the chaotic T0 = 1 problems may drift late, inside the bounds asserted before
    may_drift = float(g[f"{case}_T0"]) >= 1.0
    assert same_count or may_drift, f"{case}: {stats['nfev']} evaluations against the reference's {ref_nfev} on a non-chaotic problem"
    if not (same_count or tail_only):
        import warnings
        warnings.warn(f"test_ode_golden[{case}]: schedule drifted late (nfev {stats['nfev']} vs {ref_nfev}, first flip at attempt {first_diff}); "
                      "end pose, leading attempts and evaluation-count bounds were checked, the full-trajectory asserts were not")
    elif tail_only:
        print(f"test_ode_golden[{case}]: one attempt of difference (nfev {stats['nfev']} vs {ref_nfev}), no accept / reject flip; the first {agree} of {common} "
              "common attempts start where the reference's did (asserted), the first accepted states and the end pose were compared")


def test_pc_agent_golden(golden):
    g = golden("g7_pc.npz")
    agent = make_agent("score", "pc", 20)
    pts = torch.from_numpy(g["pts"]).cuda()
    data = {"pts": pts, "pts_center": pts.mean(dim=1)}
    with FixedPrior(agent, g["prior_noise"]):
        pred, proc = agent.pred_func(data, repeat_num=10, save_path=None, return_process=True,
                                     noise=(torch.from_numpy(g["z_langevin"]).cuda(), torch.from_numpy(g["z_predictor"]).cuda()))
    assert pred.dtype == torch.float32
    np.testing.assert_allclose(pred.cpu().numpy(), g["pred"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(proc.cpu().numpy(), g["proc"], rtol=1e-3, atol=5e-3)


def test_energy_rank_aggregate_golden(golden):
    from genpose_amd import reward, rotation
    g = golden("g8_rank.npz")
    e_agent = make_agent("energy")
    pts = torch.from_numpy(g["pts"]).cuda()
    pred = torch.from_numpy(g["pred"]).cuda()
    energy = e_agent.get_energy(data={"pts": pts, "pts_center": pts.mean(dim=1)}, pose_samples=pred, T=1e-5)
    np.testing.assert_allclose(energy.cpu().numpy(), g["energy"], rtol=5e-4, atol=5e-4 * np.abs(g["energy"]).max())
    # ranking on the golden energies: exact permutation, exact copies
    sp, se = reward.sort_poses_by_energy(pred, torch.from_numpy(g["energy"]).cuda())
    assert sp.dtype == torch.float64
    np.testing.assert_array_equal(sp.cpu().numpy(), g["sorted_pose"])
    np.testing.assert_array_equal(se.cpu().numpy(), g["sorted_energy"])
    np.testing.assert_allclose(rotation.pose9_to_RT(sp).cpu().numpy(), g["RT_sorted"], atol=1e-12)
    r = reward.rank_aggregate(pred, torch.from_numpy(g["energy"]).cuda(), ratio=0.6)
    avg_RT = rotation.quat_trans_to_RT(r["avg_pose"]).cpu().numpy()
    np.testing.assert_allclose(avg_RT, g["average_sRT"], atol=2e-6)
    # f32 poses take the same path
    r32 = reward.rank_aggregate(pred.float(), torch.from_numpy(g["energy"]).cuda(), ratio=0.6)
    np.testing.assert_allclose(r32["avg_pose"].cpu().numpy(), r["avg_pose"].cpu().numpy(), atol=1e-5)


def test_rank_ties_and_sizes():
    from genpose_amd import reward
    gen = torch.Generator().manual_seed(1)
    for B, K in [(1, 1), (3, 7), (5, 50), (2, 200)]:
        poses = torch.randn(B, K, 9, generator=gen)
        energy = torch.randint(0, 4, (B, K, 2), generator=gen).float()  # many ties -> stable order
        r = reward.rank_aggregate(poses.cuda(), energy.cuda(), ratio=0.6)
        for c in range(2):
            ref = torch.sort(energy[:, :, c], dim=1, descending=True, stable=True)
            assert torch.equal(r["order"][:, :, c].cpu().long(), ref.indices)
            assert torch.equal(r["sorted_energy"][:, :, c].cpu(), ref.values)
        avg_ref, qt = go.aggregate_sorted(go.pose9_to_RT(_ref_sorted(poses, r["order"].cpu().long())), ratio=0.6)
        np.testing.assert_allclose(r["avg_pose"].cpu().numpy()[:, 4:], qt.numpy()[:, 4:], atol=1e-5)
        dot = np.abs(np.sum(r["avg_pose"].cpu().numpy()[:, :4] * qt.numpy()[:, :4], axis=1))
        assert np.all(dot > 1 - 1e-4)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_rank_aggregate_one_launch_with_the_matrices(dtype):
    """gp_rank_aggregate_rt (ranking + aggregation + both 4x4 forms in one launch: what a tracking frame's ranking step runs) gives the
    bits of gp_rank_aggregate followed by gp_pose9_to_rt and gp_quat_trans_to_rt."""
    from genpose_amd import reward, rotation
    gen = torch.Generator().manual_seed(3)
    for B, K, sel in [(1, 1, 1), (5, 50, 30), (3, 130, 7), (64, 50, 30)]:
        poses = torch.randn(B, K, 9, generator=gen).to(dtype).cuda()
        energy = torch.randn(B, K, 2, generator=gen).cuda()
        a = reward.rank_aggregate(poses, energy, selected_num=sel)
        b = reward.rank_aggregate(poses, energy, selected_num=sel, with_rt=True)
        for k in ("sorted_poses", "sorted_energy", "order", "avg_pose"):
            assert torch.equal(a[k], b[k]), k
        assert torch.equal(b["sorted_RTs"], rotation.pose9_to_RT(a["sorted_poses"]))
        assert torch.equal(b["avg_RT"], rotation.quat_trans_to_RT(a["avg_pose"]))
        c = reward.rank_aggregate(poses, energy, with_rt=True)  # ranking only
        assert c["avg_RT"] is None and torch.equal(c["sorted_RTs"], b["sorted_RTs"])


def _ref_accepts(ref_t):
    """Accept/reject sequence of the reference run, recovered from its logged evaluation times: after a REJECTED attempt
    the next attempt restarts from the same t (its first stage time t + h'/5 lies before the rejected attempt's last
    stage time t + h), after an accepted one it starts from t + h."""
    first = ref_t[2:-1:6]   # t + h/5 of every attempt
    last = ref_t[7::6]      # t + h   of every attempt (6th evaluation)
    acc = []
    for i in range(len(first)):
        if i + 1 < len(first):
            acc.append(bool(first[i + 1] < last[i]))  # integrating DOWN in t: next attempt begins beyond t + h  <=> accepted
        else:
            acc.append(True)
    return acc


def _ref_sorted(poses, order):
    B, K, _ = poses.shape
    bi = torch.arange(B).unsqueeze(1).expand(B, K)
    out = poses[bi, order[:, :, 0]].clone()
    out[:, :, -3:] = poses[bi, order[:, :, 1]][:, :, -3:]
    return out


def test_tracking_golden(golden):
    """3-frame warm-started tracking (evaluation_tracking.py:262-337) through genpose_amd.runner.track_sequence."""
    from genpose_amd.runner import TrackingRunner
    g = golden("g9_track.npz")
    runner = TrackingRunner(make_agent("score", "ode"), make_agent("energy"), repeat_num=10, T0=0.15)
    frames = g["frames"]
    for fi in range(frames.shape[0]):
        pts = torch.from_numpy(frames[fi]).cuda()
        noise_draws = [torch.from_numpy(g[f"f{fi}_noise_draw{d}"]) for d in range(4)]
        with FixedPrior(runner.score_agent, g[f"f{fi}_prior_noise"]):
            out = runner.step(pts, model_names=["obj0", "obj1"], gt_RT=torch.from_numpy(g["gt_RT"]), noise_draws=noise_draws)
        np.testing.assert_allclose(out["init_x"].cpu().numpy(), g[f"f{fi}_init_x"], atol=2e-4 if fi else 1e-6)
        ode_close(out["pred_pose"].cpu().numpy(), g[f"f{fi}_pred"])
        np.testing.assert_allclose(out["average_sRT"].cpu().numpy(), g[f"f{fi}_avg_sRT"], atol=5e-4)


def test_likelihood_golden(golden):
    """mode='likelihood' (posenet.py:133-147, samplers.py:22-99) against the imported reference (fixture G12)."""
    g = golden("g12_likelihood.npz")
    agent = make_agent("score", "ode", None)
    pts = torch.from_numpy(g["pts"]).cuda()
    data = {"pts": pts, "pts_center": pts.mean(dim=1)}
    data["pts_feat"] = agent.net(data, mode="pts_feature")
    data["sampled_pose"] = torch.from_numpy(g["pose"]).cuda()
    probe = torch.from_numpy(g["probe"])
    saved = agent.net.prior_fn
    agent.net.prior_fn = lambda shape, **k: probe.clone()
    try:
        ll = agent.net(data, mode="likelihood")
    finally:
        agent.net.prior_fn = saved
    assert ll.dtype == torch.float64 and ll.shape == (3,)
    ref = g["log_likelihood"]
    nfev = agent.net.last_likelihood_stats["nfev"]
    # the integrand is O(1e4) bits over ~1.8e4 adaptive evaluations of a random-weight network: relative agreement
    np.testing.assert_allclose(ll.cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    assert abs(nfev - int(g["nfev"])) <= 0.05 * int(g["nfev"]), (nfev, int(g["nfev"]))
