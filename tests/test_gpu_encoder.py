"""GPU parity: fused set-abstraction kernel and the whole encoder against the oracle / golden G3 (fp32 tolerance)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go

# fp32 MFMA == fmaf chain; the oracle uses oneDNN conv + separate BN.  Features are O(1); tolerance stated here:
ENC_RTOL, ENC_ATOL = 2e-4, 2e-4


@pytest.fixture(scope="module")
def sd():
    return go.make_state_dict(0, "score")


@pytest.fixture(scope="module")
def enc(sd):
    from genpose_amd.encoder import Pointnet2EncoderHIP
    return Pointnet2EncoderHIP(sd, "cuda")


def test_encoder_golden(enc, golden):
    g = golden("g3_encoder.npz")
    pts = torch.from_numpy(g["clouds"]).cuda()
    feat, ws = enc.forward(pts, return_intermediates=True)
    for lvl in range(3):
        assert np.array_equal(ws["new_xyz"][lvl][0].cpu().numpy(), g[f"new_xyz{lvl}"])  # bit-exact sampling
        mine = ws["feat"][lvl][0].cpu().numpy()[:32].T  # point-major [n,C] -> [C, 32 points]
        np.testing.assert_allclose(mine, g[f"feat{lvl}_first32"], rtol=ENC_RTOL, atol=ENC_ATOL, err_msg=f"level {lvl}")
    np.testing.assert_allclose(feat.cpu().numpy(), g["feat"], rtol=ENC_RTOL, atol=ENC_ATOL)


def test_encoder_vs_oracle_batches(enc, sd):
    from genpose_amd import synth
    for B, start in [(1, 0), (5, 100), (16, 40)]:
        pts = synth.make_batch(B, start)
        ref = go.encoder_forward(sd, torch.from_numpy(pts)).numpy()
        got = enc.forward(torch.from_numpy(pts).cuda()).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=ENC_RTOL, atol=ENC_ATOL)


@pytest.mark.parametrize("arith", ["A", "B", "C"])
def test_encoder_under_every_distance_convention(sd, golden, arith):
    """The contraction convention of the grouping operators' squared distances (include/genpose_hip.h GP_ARITH_*) as an encoder / agent
    field: under each of the three the centres and neighbourhoods of every level equal the oracle's under the SAME convention bit for
    bit - on clouds where the conventions disagree (golden clouds: grid ties, tiled duplicates; synthetic clouds 23 and 74 of the
    2048-cloud sensitivity run, profiles/r5_arith_sensitivity.txt) - and the features follow to the fp32 tolerance."""
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.encoder import Pointnet2EncoderHIP
    from genpose_amd.posenet_agent import PoseNet
    from oracle import pn2_oracle as ops
    all_synth = synth.make_batch(75)
    pts = np.concatenate([golden("g3_encoder.npz")["clouds"], all_synth[[23, 74, 5]]])
    enc_a = Pointnet2EncoderHIP(sd, "cuda", arith=arith)
    feat, ws = enc_a.forward(torch.from_numpy(pts).cuda(), return_intermediates=True)
    with ops.use_arith(arith):
        ref, inter = go.encoder_forward(sd, torch.from_numpy(pts), return_intermediates=True)
    for lvl in range(3):
        assert np.array_equal(ws["fps_idx"][lvl].cpu().numpy(), inter[lvl]["fps_idx"]), f"{arith}: FPS level {lvl}"
        assert np.array_equal(ws["new_xyz"][lvl].cpu().numpy(), inter[lvl]["new_xyz"])
        for s in range(2):
            assert np.array_equal(ws["bq"][lvl][s].cpu().numpy(), inter[lvl][f"bq_idx{s}"]), f"{arith}: ball query level {lvl} scale {s}"
    np.testing.assert_allclose(feat.cpu().numpy(), ref.numpy(), rtol=ENC_RTOL, atol=ENC_ATOL)
    # the conventions really select differently on these clouds (level-0 picks of the two synthetic clouds)
    if arith != ops.DEFAULT_ARITH:
        dflt = ops.furthest_point_sampling(pts, 512)[0]
        assert not np.array_equal(dflt, inter[0]["fps_idx"])
    # and through the agent: cfg.dist_arith
    agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"], dist_arith=arith))
    agent.load_state_dict(sd)
    assert agent.net.pts_encoder.arith == ops.ARITH_CODES[arith]
    data = {"pts": torch.from_numpy(pts).cuda()}
    got = agent.net.extract_pts_feature(data) if hasattr(agent.net, "extract_pts_feature") else None
    if got is not None:
        assert torch.equal(got, feat)


def test_encoder_is_deterministic_and_batch_independent(enc):
    from genpose_amd import synth
    pts = torch.from_numpy(synth.make_batch(8, 7)).cuda()
    a = enc.forward(pts).clone()
    b = enc.forward(pts).clone()
    assert torch.equal(a, b)
    c = enc.forward(pts[2:5].contiguous())
    assert torch.equal(a[2:5], c)  # clouds are independent units (sharding property)


def test_energy_weights_encoder(golden):
    from genpose_amd.encoder import Pointnet2EncoderHIP
    sde = go.make_state_dict(0, "energy")
    g = golden("g3_encoder.npz")
    pts = torch.from_numpy(g["clouds"][:2])
    ref = go.encoder_forward(sde, pts).numpy()
    got = Pointnet2EncoderHIP(sde, "cuda").forward(pts.cuda()).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=ENC_RTOL, atol=ENC_ATOL)


def test_grouping_ticket_dies_with_the_workspace_contents():
    """The score agent leaves a ticket for its centres / neighbourhoods in the data dict; the energy agent honours it only while the
    workspace still holds THAT grouping for THAT tensor.  Every writer of the grouping buffers (a plain forward of another cloud batch of the
    same shape, sample_centres) and every in-place edit of the clouds must invalidate it - otherwise the energy encoder would silently run
    with another cloud's centres."""
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    sa = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=4))
    sa.load_state_dict(go.make_state_dict(0, "score"))
    ea = PoseNet(get_config(posenet_mode="energy"))
    ea.load_state_dict(go.make_state_dict(0, "energy"))
    B, K = 4, 5
    pts = torch.from_numpy(synth.make_batch(B, start=900)).cuda()
    other = torch.from_numpy(synth.make_batch(B, start=950)).cuda()
    poses = torch.randn(B, K, 9, generator=torch.Generator().manual_seed(0)).cuda()
    want = ea.get_energy({"pts": pts, "pts_center": pts.mean(dim=1)}, poses, T=1e-5).clone()  # fresh dict: no ticket, own grouping
    enc_s, enc_e = sa.net.pts_encoder, ea.net.pts_encoder

    def ticketed():
        data = {"pts": pts, "pts_center": pts.mean(dim=1)}
        sa.pred_func(data, K, save_path=None)
        assert enc_e.ticket_valid(data["_grouping"], pts, enc_e.grouping_key())
        return data

    # the intended use: same dict, same tensor, workspace untouched -> taken over, same bits
    data = ticketed()
    assert torch.equal(ea.get_energy(data, poses, T=1e-5), want)
    # a plain forward of OTHER clouds of the same shape overwrites the workspace -> the ticket is dead, the result is still right
    data = ticketed()
    enc_s(other)
    assert not enc_e.ticket_valid(data["_grouping"], pts, enc_e.grouping_key())
    assert torch.equal(ea.get_energy(data, poses, T=1e-5), want)
    # so does sample_centres alone
    data = ticketed()
    enc_s.sample_centres(other)
    assert not enc_e.ticket_valid(data["_grouping"], pts, enc_e.grouping_key())
    assert torch.equal(ea.get_energy(data, poses, T=1e-5), want)
    # an in-place edit of the clouds, and a different tensor object holding the same values
    data = ticketed()
    pts.add_(0.0)
    assert not enc_e.ticket_valid(data["_grouping"], pts, enc_e.grouping_key())
    data = ticketed()
    assert not enc_e.ticket_valid(data["_grouping"], pts.clone(), enc_e.grouping_key())


@pytest.mark.parametrize("params", ["dense", "lighter"])
def test_other_encoder_configurations(golden, params):
    """--pointnet2_params dense | lighter (pointnet2.py:47-78): neighbourhoods of 8 and 64 samples, four single-scale grouping levels and
    a 512 -> 512 -> 1024 GroupAll level.  Against the fixture captured from the imported reference (G16), then against the oracle at 5 and
    64 clouds with bit-exact centres and neighbourhoods on every level; and through the agent (`cfg.pointnet2_params`)."""
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.encoder import Pointnet2EncoderHIP
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.weights import ENCODER_CFGS
    cfg = ENCODER_CFGS[params]
    nlev = sum(1 for n in cfg["npoints"] if n is not None)
    sd = go.make_state_dict(0, "score", params)
    enc = Pointnet2EncoderHIP(sd, "cuda", params)
    g = golden(f"g16_encoder_{params}.npz")
    feat, ws = enc.forward(torch.from_numpy(g["clouds"]).cuda(), return_intermediates=True)
    for lvl in range(nlev):
        assert np.array_equal(ws["new_xyz"][lvl][0].cpu().numpy(), g[f"new_xyz{lvl}"])
        mine = ws["feat"][lvl][0].cpu().numpy()[:32].T
        np.testing.assert_allclose(mine, g[f"feat{lvl}_first32"], rtol=ENC_RTOL, atol=ENC_ATOL, err_msg=f"{params} level {lvl}")
    np.testing.assert_allclose(feat.cpu().numpy(), g["feat"], rtol=ENC_RTOL, atol=ENC_ATOL)
    for B, start in [(5, 300), (64, 1300)]:
        pts = synth.make_batch(B, start)
        got, ws = enc.forward(torch.from_numpy(pts).cuda(), return_intermediates=True)
        got = got.cpu().numpy()
        for s in range(0, B, 16):
            ref, inter = go.encoder_forward(sd, torch.from_numpy(pts[s:s + 16]), cfg=cfg, return_intermediates=True)
            np.testing.assert_allclose(got[s:s + 16], ref.numpy(), rtol=ENC_RTOL, atol=ENC_ATOL, err_msg=f"{params} B={B} clouds {s}..")
            for lvl in range(nlev):
                assert np.array_equal(ws["fps_idx"][lvl][s:s + 16].cpu().numpy(), inter[lvl]["fps_idx"]), (params, B, lvl)
                for i in range(len(cfg["radii"][lvl])):
                    assert np.array_equal(ws["bq"][lvl][i][s:s + 16].cpu().numpy(), inter[lvl][f"bq_idx{i}"]), (params, B, lvl, i)
    agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"], pointnet2_params=params))
    agent.load_state_dict(sd)
    pts = torch.from_numpy(synth.make_batch(3, 40)).cuda()
    pred = agent.pred_func({"pts": pts, "pts_center": pts.mean(dim=1)}, repeat_num=4, save_path=None, T0=0.3)
    assert pred.shape == (3, 4, 9) and torch.isfinite(pred).all()
