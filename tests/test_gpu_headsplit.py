"""GPU: the head-split launch plan of the latency regime (GP_PLAN_HEADSPLIT, round 5): three workgroups per 16-row tile, each
recomputing pose_encoder and evaluating ONE of the three fusion tails (scorenet.py:178-222).  Against the plain 16-row tiles (the same
components must come out bit for bit: same MFMA sequence per accumulator, same combine order), against the oracle, and through the
samplers that pick it on their own (a tracking frame, BASELINE configs[0], small agent calls)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go

HS = 0x100


@pytest.fixture(scope="module")
def nets():
    from genpose_amd.scorenet import ScoreNetHIP
    sd = go.make_state_dict(0, "score")
    return sd, ScoreNetHIP(sd, "cuda")


def _inputs(B, K, n, seed=0):
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(B, 1024, generator=g) * 0.5
    centre = torch.randn(B, 3, generator=g) * 0.1
    x0 = torch.randn(B * K, 9, generator=g) * 50.0
    z1, z2 = torch.randn(n, B * K, 9, generator=g), torch.randn(n, B * K, 9, generator=g)
    return feat, centre, x0, z1, z2


def test_plan_rule():
    """Picked while three workgroups per tile still get a CU each (tiles x 3 <= 256), for the score model only; never by the entry points
    whose partial-sum contract predates it."""
    from genpose_amd import _lib
    lib = _lib.lib()
    assert lib.gp_plan_headsplit_pays(85) == 1 and lib.gp_plan_headsplit_pays(86) == 0
    assert lib.gp_rk45_plan_rows(0, 1, 5, 50) == 16 | HS      # a tracking frame: 250 rows = 16 tiles
    assert lib.gp_rk45_plan_rows(0, 1, 1, 10) == 16 | HS      # BASELINE configs[0]: one tile
    assert lib.gp_rk45_plan_rows(0, 1, 27, 50) == 16 | HS     # 1350 rows = 85 tiles
    assert lib.gp_rk45_plan_rows(0, 1, 28, 50) == 16          # 88 tiles: plain tiles again
    assert lib.gp_rk45_plan_rows(0, 1, 256, 50) in (32, 64, 48 | 0x200)  # (256 CUs: the shared-chunk plan, tests/test_gpu_shared_plan.py)
    assert lib.gp_rk45_plan_rows_unshared(0, 1, 256, 50) in (32, 64)
    assert lib.gp_rk45_plan_rows(1, 1, 5, 50) == 16 and lib.gp_rk45_plan_rows(2, 1, 5, 50) == 16  # forward + backward right-hand sides: tiles
    t, n = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.gp_pc_layout(0, 0, 1, 5, 50, ctypes.byref(t), ctypes.byref(n)) == 0 and t.value == 16 | HS and n.value == 21 * 250
    assert lib.gp_pc_layout(0, 16, 1, 5, 50, ctypes.byref(t), ctypes.byref(n)) == 0 and t.value == 16 and n.value == 16
    assert lib.gp_pc_layout(1, 16 | HS, 1, 5, 50, ctypes.byref(t), ctypes.byref(n)) == -1  # the energy model's score: no head-split
    assert lib.gp_pc_tile_rows(1, 5, 50) == 16  # the legacy entry points stay on whole tiles


@pytest.mark.parametrize("B,K", [(1, 10), (3, 10), (5, 50), (2, 43)])
def test_pc_sampler_split_vs_tiles_vs_oracle(nets, B, K):
    from genpose_amd.samplers import PCSampler
    sd, net = nets
    n = 12
    feat, centre, x0, z1, z2 = _inputs(B, K, n, seed=B * 100 + K)
    cvec = net.cloud_embed(feat.cuda())
    outs = {}
    for plan in (16, 16 | HS):
        smp = PCSampler(net, B, K, n, "cuda", record_traj=True, tile=plan)
        assert smp.plan == plan and smp.tile == 16 and smp.hsplit == (3 if plan & HS else 1)
        # launch 0 alone: the first score evaluation - component for component the same bits under both plans
        smp.cvec.copy_(cvec), smp.centre.copy_(centre.cuda()), smp.x.copy_(x0.cuda()), smp.z1.copy_(z1.cuda()), smp.z2.copy_(z2.cuda())
        smp.launch_step(0)
        torch.cuda.synchronize()
        first = smp.score.clone()
        xs, mean_x = smp.run(cvec, centre.cuda(), x0.cuda(), z1.cuda(), z2.cuda())
        again = smp.run(cvec, centre.cuda(), x0.cuda(), z1.cuda(), z2.cuda())[1].clone()  # the captured graph replays to the same bits
        assert torch.equal(again, mean_x)
        outs[plan] = (first, mean_x.clone(), xs.clone())
    assert torch.equal(outs[16][0], outs[16 | HS][0])
    # the step size takes the batch mean of the row norms, put together in a different order: round-off through the recursion
    scale = float(outs[16][1].abs().max())
    np.testing.assert_allclose(outs[16 | HS][1].cpu().numpy(), outs[16][1].cpu().numpy(), rtol=0, atol=2e-5 * scale)
    feat_r, cen_r = feat.repeat_interleave(K, 0), centre.repeat_interleave(K, 0)
    _, ref = go.pc_sampler(lambda x, t: go.score_forward(sd, feat_r, x, t), x0, cen_r, n, z1, z2)
    np.testing.assert_allclose(outs[16 | HS][1].cpu().numpy(), ref.numpy(), rtol=0, atol=1e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("B,K,T0", [(1, 10, 0.55), (5, 50, 0.15), (3, 7, 1.0)])
def test_ode_sampler_split_vs_tiles_vs_oracle(nets, B, K, T0):
    """Every stage of an attempt is its own launch under the head-split plan (the fused attempt kernel of the plain tiles cannot be: a
    stage reads all nine components of the previous one).  Same evaluation count as the plain tiles, poses equal to f64 round-off
    (the error-norm partial sums are added in a different order), the oracle's solve within the ODE tolerance."""
    from genpose_amd.samplers import ODESampler
    sd, net = nets
    feat, centre, x0, _, _ = _inputs(B, K, 1, seed=7 * B + K)
    x0 = x0 / 50.0 * float(go.ve_sigma(T0))
    cvec = net.cloud_embed(feat.cuda())
    res = {}
    for plan in (16, 16 | HS):
        smp = ODESampler(net, B, K, "cuda", tile=plan)
        assert smp.plan == plan and smp.partials.shape == (3, smp.nblocks * smp.hsplit)
        _, x = smp.run(cvec, centre.cuda(), x0.cuda(), T0)
        _, x2 = smp.run(cvec, centre.cuda(), x0.cuda(), T0)
        assert torch.equal(x, x2)
        res[plan] = (x.clone(), int(smp.last_stats["nfev"]), smp.last_stats["log_err"].copy())
    assert res[16][1] == res[16 | HS][1]
    np.testing.assert_allclose(res[16 | HS][2], res[16][2], rtol=1e-9)
    np.testing.assert_allclose(res[16 | HS][0].cpu().numpy(), res[16][0].cpu().numpy(), rtol=0, atol=1e-9 * float(res[16][0].abs().max()))
    feat_r, cen_r = feat.repeat_interleave(K, 0), centre.repeat_interleave(K, 0)
    _, ref, nfev = go.ode_sampler(lambda x, t: go.score_forward(sd, feat_r, x, t), x0, cen_r, T0)
    assert abs(res[16 | HS][1] - nfev) <= 6
    got = res[16 | HS][0].cpu().numpy()
    np.testing.assert_allclose(got[:, :6], ref.numpy()[:, :6], rtol=0, atol=2e-3)
    np.testing.assert_allclose(got[:, 6:], ref.numpy()[:, 6:], rtol=0, atol=5e-4 * max(1.0, float(ref[:, 6:].abs().max())))


def test_ragged_groups_split_vs_tiles(nets):
    """The multi-sequence tracker's solver (ragged groups, one step controller per group) under both plans: per group the same
    evaluation count and poses to f64 round-off; groups whose sizes change between solves reuse the solver."""
    from genpose_amd.samplers import ODESampler
    sd, net = nets
    K, T0 = 8, 0.15
    feat, centre, x0, _, _ = _inputs(9, K, 1, seed=3)
    x0 = x0 / 50.0 * float(go.ve_sigma(T0))
    cvec = net.cloud_embed(feat.cuda())
    out = {}
    for plan in (16, 16 | HS):
        smp = ODESampler(net, 12, K, "cuda", group_clouds=[4, 4, 4], tile=plan)
        assert smp.plan == plan
        rec = []
        for groups in ([2, 3, 1], [4, 1, 4]):
            nb = sum(groups)
            smp.set_groups(groups)
            _, x = smp.run(cvec[:nb], centre[:nb].cuda(), x0[: nb * K].cuda(), T0)
            rec.append((x.clone(), [int(s_["nfev"]) for s_ in smp.group_stats]))
        out[plan] = rec
    auto = ODESampler(net, 12, K, "cuda", group_clouds=[4, 4, 4])
    assert auto.plan == 16 | HS  # 9 tiles at capacity: the latency regime
    for a, b in zip(out[16], out[16 | HS]):
        assert a[1] == b[1]
        np.testing.assert_allclose(b[0].cpu().numpy(), a[0].cpu().numpy(), rtol=0, atol=1e-9 * float(a[0].abs().max()))


def test_agent_calls_take_the_plan_and_a_sharded_batch_does_not(nets):
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd import synth
    sd, _ = nets
    pts = torch.from_numpy(synth.make_batch(2, start=11)).cuda()
    for sampler, steps in (("pc", 8), ("ode", None)):
        agent = PoseNet(get_config(posenet_mode="score", sampler_mode=[sampler], sampling_steps=steps))
        agent.load_state_dict(sd)
        pred = agent.pred_func({"pts": pts, "pts_center": pts.mean(dim=1)}, repeat_num=10, save_path=None, T0=0.55)
        assert torch.isfinite(pred).all() and agent.net.last_sampler.hsplit == 3
    # coupling (a batch sharded over ranks): the per-step sums between the launches run over whole-tile partials
    import torch.distributed as dist
    import os, socket
    if not dist.is_initialized():
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=0, world_size=1)
        made = True
    else:
        made = False
    try:
        from genpose_amd.samplers import PCSampler
        from genpose_amd.scorenet import ScoreNetHIP
        smp = PCSampler(ScoreNetHIP(sd, "cuda"), 2, 10, 4, "cuda", coupling_group=dist.group.WORLD)
        assert smp.plan == 16 and smp.hsplit == 1
    finally:
        if made:
            dist.destroy_process_group()


def test_split_plan_under_contention_is_bit_stable(nets):
    """Three workgroups serve a tile: each reads the tile's whole state and score and writes a part.  Nothing a launch reads may be
    written by the same launch (a sibling dispatched late - other streams own the CUs - would read what another has already written;
    an in-place version of this plan passed every quiet test and failed next to a busy encoder stream).  Every step therefore keeps
    its own copies (PcArgs, csrc/scorenet.hip).  Here: the sampler's graph replayed while an encoder pass of 256 clouds hammers the chip
    on another stream, 20 times - every replay must give the bits of the undisturbed run; the RK45 driver likewise."""
    from genpose_amd import synth
    from genpose_amd.encoder import Pointnet2EncoderHIP
    from genpose_amd.samplers import ODESampler, PCSampler
    sd, net = nets
    B, K, n = 6, 50, 25
    feat, centre, x0, z1, z2 = _inputs(B, K, n, seed=99)
    cvec = net.cloud_embed(feat.cuda())
    smp = PCSampler(net, B, K, n, "cuda", tile=16 | HS)
    quiet = smp.run(cvec, centre.cuda(), x0.cuda(), z1.cuda(), z2.cuda())[1].clone()
    ode = ODESampler(net, B, K, "cuda", tile=16 | HS)
    y0 = (x0 / 50.0 * float(go.ve_sigma(0.15))).cuda()
    quiet_ode = ode.run(cvec, centre.cuda(), y0, 0.15)[1].clone()
    enc = Pointnet2EncoderHIP(sd, "cuda")
    big = torch.from_numpy(synth.make_batch(256, start=5000)).cuda()
    side = torch.cuda.Stream()
    enc.forward(big)
    torch.cuda.synchronize()
    for rep in range(20):
        with torch.cuda.stream(side):
            enc.forward(big)  # ~4 ms of kernels that fill every CU
        got = smp.run(cvec, centre.cuda(), x0.cuda(), z1.cuda(), z2.cuda())[1]
        assert torch.equal(got, quiet), f"PC replay {rep} beside a busy stream"
        if rep % 5 == 0:
            with torch.cuda.stream(side):
                enc.forward(big)
            assert torch.equal(ode.run(cvec, centre.cuda(), y0, 0.15)[1], quiet_ode), f"ODE solve {rep} beside a busy stream"
    torch.cuda.synchronize()
