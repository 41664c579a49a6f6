"""GPU edge cases: empty and degenerate inputs, argument validation at the C ABI, extreme candidate counts (tiles that span many
clouds / a single row), each against the oracle where there is something to compare."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go
from oracle import pn2_oracle as ops


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_c_abi_rejects_bad_arguments_and_accepts_empty_batches():
    from genpose_amd import _lib
    l = _lib.lib()
    _lib.check_device()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    xyz = torch.randn(2, 64, 3, device="cuda")
    idx = torch.full((2, 8), -7, dtype=torch.int32, device="cuda")
    temp = torch.full((2, 64), 1e10, device="cuda")
    assert l.gp_furthest_point_sampling(2, 64, 8, None, _p(temp), _p(idx), st) == -1          # GP_EINVAL: null pointer
    assert l.gp_furthest_point_sampling(2, 0, 8, _p(xyz), _p(temp), _p(idx), st) == -1         # n = 0
    assert l.gp_furthest_point_sampling(2, 64, 65, _p(xyz), _p(temp), _p(idx), st) in (-1, 0)  # m > n: refused or clamped, never a crash
    fresh = torch.full((2, 8), -7, dtype=torch.int32, device="cuda")
    assert l.gp_furthest_point_sampling(0, 64, 8, _p(xyz), _p(temp), _p(fresh), st) == 0       # empty batch: no-op
    torch.cuda.synchronize()
    assert bool((fresh == -7).all())                                                            # ... that writes nothing
    assert l.gp_ball_query(0, 64, 8, ctypes.c_float(0.1), 4, _p(xyz), _p(xyz), _p(idx), st) == 0
    assert l.gp_ball_query(2, 64, 8, ctypes.c_float(0.1), 0, _p(xyz), _p(xyz), _p(idx), st) == -1  # nsample = 0
    net = _lib.GpScoreNet()
    assert l.gp_score_eval(0, 5, ctypes.byref(net), _p(xyz), _p(xyz), _p(xyz), _p(xyz), 0, _p(xyz), st) == 0   # no rows
    assert l.gp_score_eval(1, 0, ctypes.byref(net), _p(xyz), _p(xyz), _p(xyz), _p(xyz), 0, _p(xyz), st) == -1  # k = 0
    assert l.gp_score_eval(1, 5, ctypes.byref(net), _p(xyz), _p(xyz), _p(xyz), _p(xyz), 7, _p(xyz), st) == -1  # unknown mode
    assert l.gp_pc_tile_rows(2, 3, 10) == -1                                                                   # no tile divides a batch
    assert l.gp_rank_aggregate(0, 50, 30, 0, _p(xyz), _p(xyz), _p(xyz), _p(xyz), _p(idx), _p(xyz), st) == 0
    # round-6 entry points: the one-launch ranking with the 4x4 forms, the RK45 plan queries
    assert l.gp_rank_aggregate_rt(0, 50, 30, 0, _p(xyz), _p(xyz), _p(xyz), _p(xyz), _p(idx), _p(xyz), _p(xyz), _p(xyz), st) == 0   # empty batch
    assert l.gp_rank_aggregate_rt(2, 50, 30, 0, _p(xyz), _p(xyz), None, None, None, None, None, _p(xyz), st) == -1  # an aggregated 4x4 without the aggregate
    assert l.gp_rank_aggregate_rt(2, 50, 51, 0, _p(xyz), _p(xyz), None, None, None, _p(xyz), None, None, st) == -1   # more selected than candidates
    assert l.gp_rank_aggregate_rt(2, 0, 0, 0, _p(xyz), _p(xyz), None, None, None, None, None, None, st) == -1        # k = 0
    assert l.gp_rk45_partials_count(0, 0, 0, 5, 50) == -1 and l.gp_rk45_partials_count(3, 0, 1, 5, 50) == -1           # no group / unknown model
    assert l.gp_rk45_partials_count(0, 48, 1, 256, 50) == -1 and l.gp_rk45_partials_count(0, 0x200, 1, 256, 50) == -1  # 48 rows only under the shared plan; no tile size
    assert l.gp_rk45_plan_rows(0, 0, 5, 50) == -1 and l.gp_rk45_plan_rows_unshared(0, 1, 0, 50) == -1
    # a plan that does not serve the shape is refused by the driver itself (no launch): 48 | GP_PLAN_SHARED on 64 clouds, on two groups, for the energy model
    st8 = torch.zeros(2 * int(l.gp_rk45_state_bytes()), dtype=torch.uint8, device="cuda")
    buf = torch.zeros(64 * 50 * 9 * 8, dtype=torch.float64, device="cuda")
    args = lambda model, plan, groups, per: (model, plan, None, 3, groups, per, 50, ctypes.byref(net), _p(xyz), _p(xyz), _p(xyz), _p(st8), _p(buf), _p(buf), _p(buf),
                                             _p(buf), None, 0, ctypes.c_double(0.55), ctypes.c_double(1e-5), ctypes.c_double(1e-5), ctypes.c_double(1e-5),
                                             ctypes.c_double(0.0), 1, 0, _p(buf), None, 0, st)
    assert l.gp_rk45_phase_model(*args(0, 48 | 0x200, 1, 64)) == -1
    assert l.gp_rk45_phase_model(*args(0, 48 | 0x200, 2, 128)) == -1
    assert l.gp_rk45_phase_model(*args(1, 48 | 0x200, 1, 256)) == -1


@pytest.mark.parametrize("kind", ["all_identical", "two_points", "collinear_grid"])
def test_degenerate_clouds_bit_exact_sampling_and_finite_features(kind):
    """Clouds in which almost every distance ties: the FPS tie rule and the ball query's first-hit padding decide
    everything.  Indices bit-exact vs the oracle; the encoder's features finite and within tolerance."""
    from genpose_amd.encoder import Pointnet2EncoderHIP
    n = 1024
    if kind == "all_identical":
        cloud = np.tile(np.array([[0.1, -0.2, 0.8]], dtype=np.float32), (n, 1))
    elif kind == "two_points":
        cloud = np.where((np.arange(n) % 2 == 0)[:, None], np.float32([0.0, 0.0, 0.7]), np.float32([0.03, 0.0, 0.7])).astype(np.float32)
    else:
        g = (np.arange(n) % 32).astype(np.float32) * 0.005
        cloud = np.stack([g, np.zeros(n, np.float32), np.full(n, 0.9, np.float32)], axis=1)
    pts = np.stack([cloud, cloud[::-1].copy()])
    sd = go.make_state_dict(0, "score")
    enc = Pointnet2EncoderHIP(sd, "cuda")
    feat, ws = enc.forward(torch.from_numpy(pts).cuda(), return_intermediates=True)
    ref, inter = go.encoder_forward(sd, torch.from_numpy(pts), return_intermediates=True)
    for lvl in range(3):
        assert np.array_equal(ws["fps_idx"][lvl].cpu().numpy(), inter[lvl]["fps_idx"]), f"{kind}: fps level {lvl}"
        for i in range(2):
            assert np.array_equal(ws["bq"][lvl][i].cpu().numpy(), inter[lvl][f"bq_idx{i}"]), f"{kind}: ball query level {lvl} scale {i}"
    assert torch.isfinite(feat).all()
    np.testing.assert_allclose(feat.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("B,K", [(1, 1), (20, 3), (2, 200), (3, 7)])
def test_extreme_candidate_counts(B, K):
    """One row in all (B = K = 1); 16-row tiles that span six clouds (K = 3: the head epilogue's operands come from global memory, not
    from the two staged clouds); one cloud spanning many tiles (K = 200); a ragged last tile (21 rows).  Score, energy and a short PC
    run against the oracle."""
    from genpose_amd.samplers import PCSampler
    from genpose_amd.scorenet import ScoreNetHIP
    gen = torch.Generator().manual_seed(100 * B + K)
    pf = torch.randn(B, 1024, generator=gen).abs()
    pose = torch.randn(B * K, 9, generator=gen)
    for mode, fwd in (("score", go.score_forward), ("energy", go.energy_forward)):
        sd = go.make_state_dict(0, mode)
        net = ScoreNetHIP(sd, "cuda")
        t = 0.3
        ref = fwd(sd, pf.repeat_interleave(K, 0), pose, torch.ones(B * K, 1) * t).numpy()
        cvec = net.cloud_embed(pf.cuda())
        tvec = net.time_embed(torch.tensor([t], device="cuda"))
        sigma = torch.tensor([0.01 * 5000.0 ** t], device="cuda")
        got = net.evaluate(cvec, K, pose.cuda(), tvec[0], sigma, mode).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())
    sd = go.make_state_dict(0, "score")
    net = ScoreNetHIP(sd, "cuda")
    n = 5
    centre = torch.randn(B, 3, generator=gen) * 0.3
    x0 = torch.randn(B * K, 9, generator=gen) * 50.0
    z1, z2 = torch.randn(n, B * K, 9, generator=gen), torch.randn(n, B * K, 9, generator=gen)
    fr = pf.repeat_interleave(K, 0)
    _, ref = go.pc_sampler(lambda x, tt: go.score_forward(sd, fr, x, tt), x0, centre.repeat_interleave(K, 0), n, z1, z2)
    _, got = PCSampler(net, B, K, n, "cuda").run(net.cloud_embed(pf.cuda()), centre.cuda(), x0.cuda(), z1.cuda(), z2.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()))


def test_empty_frame_and_single_object_through_the_runners():
    """A tracking step in which one sequence has no frame and another a single object; a single-frame batch of one cloud."""
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.runner import MultiSequenceTracker, SingleFrameRunner
    sa = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"]))
    sa.load_state_dict(go.make_state_dict(0, "score"))
    ea = PoseNet(get_config(posenet_mode="energy"))
    ea.load_state_dict(go.make_state_dict(0, "energy"))
    pts = torch.from_numpy(synth.make_batch(1, start=77)).cuda()
    trk = MultiSequenceTracker(sa, ea, n_sequences=3, repeat_num=6, T0=0.15)
    out = trk.step([None, (pts, ["only"], torch.eye(4).unsqueeze(0)), None])
    assert out[0] is None and out[2] is None and out[1]["pred_pose"].shape == (1, 6, 9) and torch.isfinite(out[1]["average_sRT"]).all()
    assert trk.step([None, None, None]) == [None, None, None]
    res = SingleFrameRunner(sa, ea, repeat_num=6, T0=0.3, batch_size=4).infer(pts.cpu().numpy())
    assert res["average_sRT"].shape == (1, 4, 4) and np.isfinite(res["average_sRT"]).all()


def test_legacy_pc_step_entry_points_keep_their_partials_contract():
    """gp_pc_step / gp_pc_step_grouped predate the launch plans: a C caller sizes `partials` as nsteps x ceil(R / gp_score_tile_rows(R))
    (include/genpose_hip.h).  At 32 050 rows the automatic plan is the chain form, which writes one partial per WAVE (1 004 of them);
    the legacy entry points must stay in the tile form and inside the buffer their contract states - a guard band behind it stays
    untouched - and give what the planned launch chain gives in that form."""
    from genpose_amd import _lib
    from genpose_amd.samplers import PCSampler, pc_schedule
    from genpose_amd.scorenet import ScoreNetHIP
    l = _lib.lib()
    B, K, n = 641, 50, 2
    R = B * K
    tile = l.gp_score_tile_rows(R)
    t_out, n_out = ctypes.c_int(0), ctypes.c_int(0)
    assert l.gp_pc_layout(0, 0, 1, B, K, ctypes.byref(t_out), ctypes.byref(n_out)) == 0 and t_out.value == 128  # what the plan would take
    nparts = (R + tile - 1) // tile
    assert n_out.value > nparts
    net = ScoreNetHIP(go.make_state_dict(0, "score"), "cuda")
    gen = torch.Generator().manual_seed(9)
    cvec = torch.randn(B, 768, generator=gen).cuda()
    centre = (torch.randn(B, 3, generator=gen) * 0.3).cuda()
    x0 = (torch.randn(R, 9, generator=gen) * 50.0).cuda()
    z1, z2 = torch.randn(n, R, 9, generator=gen).cuda(), torch.randn(n, R, 9, generator=gen).cuda()
    ts, sched = pc_schedule(n)
    sched = sched.cuda()
    tvec = net.time_embed(ts.cuda())
    x, mean_x, score = x0.clone(), torch.empty(R, 9, device="cuda"), torch.empty(R, 9, device="cuda")
    guard = 64
    partials = torch.full((n * nparts + guard,), -123.0, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(n + 1):
        assert l.gp_pc_step(B, K, i, n, net.w.ref(), _p(cvec), _p(tvec), _p(sched), _p(z1), _p(z2), _p(centre), _p(x), _p(mean_x), _p(score),
                            _p(partials), None, st) == 0
    torch.cuda.synchronize()
    assert bool((partials[n * nparts:] == -123.0).all()) and bool((partials[: n * nparts] != -123.0).all())
    _, want = PCSampler(net, B, K, n, "cuda", tile=tile, use_graph=False).run(cvec, centre, x0, z1, z2)
    assert torch.equal(mean_x, want)
