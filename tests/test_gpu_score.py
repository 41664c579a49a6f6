"""GPU parity: score / energy networks (G4/G5) and the PC sampler (G7) against golden vectors and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go

# f_theta is O(1), divided by sigma(t) (down to 1e-2): relative tolerance on the score, stated here
NET_RTOL = 2e-4


@pytest.fixture(scope="module")
def nets():
    from genpose_amd.scorenet import ScoreNetHIP
    return ScoreNetHIP(go.make_state_dict(0, "score"), "cuda"), ScoreNetHIP(go.make_state_dict(0, "energy"), "cuda")


def test_score_energy_golden(nets, golden):
    g = golden("g4_g5_nets.npz")
    snet, enet = nets
    pf, pose = torch.from_numpy(g["pts_feat"]).cuda(), torch.from_numpy(g["pose"]).cuda()
    for i, t in enumerate(g["t"]):
        tt = torch.ones(8, 1, device="cuda") * float(t)
        s = snet.forward_rows(pf, pose, tt, "score").cpu().numpy()
        e = enet.forward_rows(pf, pose, tt, "energy").cpu().numpy()
        np.testing.assert_allclose(s, g[f"score_{i}"], rtol=NET_RTOL, atol=NET_RTOL * np.abs(g[f"score_{i}"]).max())
        np.testing.assert_allclose(e, g[f"energy_{i}"], rtol=NET_RTOL, atol=NET_RTOL * np.abs(g[f"energy_{i}"]).max())


def test_score_rows_share_cloud_embedding(nets):
    """cvec hoisting: B clouds x K candidates with one embedding per cloud == per-row evaluation; ragged tile sizes."""
    snet, _ = nets
    sd = go.make_state_dict(0, "score")
    gen = torch.Generator().manual_seed(3)
    for B, K in [(1, 1), (3, 50), (7, 10), (2, 33)]:
        pf = torch.randn(B, 1024, generator=gen).abs()
        pose = torch.randn(B * K, 9, generator=gen)
        t = 0.3
        ref = go.score_forward(sd, pf.repeat_interleave(K, 0), pose, torch.ones(B * K, 1) * t).numpy()
        cvec = snet.cloud_embed(pf.cuda())
        tvec = snet.time_embed(torch.tensor([t], device="cuda"))
        sigma = torch.tensor([0.01 * 5000.0 ** t], device="cuda")
        got = snet.evaluate(cvec, K, pose.cuda(), tvec[0], sigma, "score").cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=NET_RTOL, atol=NET_RTOL * np.abs(ref).max())


@pytest.mark.parametrize("use_graph", [False, True])
def test_pc_sampler_golden(nets, golden, use_graph):
    from genpose_amd.encoder import Pointnet2EncoderHIP
    from genpose_amd.samplers import PCSampler
    g = golden("g7_pc.npz")
    snet, _ = nets
    sd = go.make_state_dict(0, "score")
    pts = torch.from_numpy(g["pts"]).cuda()
    feat = Pointnet2EncoderHIP(sd, "cuda").forward(pts)
    cvec = snet.cloud_embed(feat)
    B, K, n = 2, 10, 20
    smp = PCSampler(snet, B, K, n, "cuda", use_graph=use_graph, record_traj=True)
    init_x = torch.from_numpy(g["prior_noise"]) * 50.0
    for _ in range(2):  # second call replays the captured graph
        xs, mean_x = smp.run(cvec, pts.mean(dim=1), init_x.cuda(), torch.from_numpy(g["z_langevin"]).cuda(),
                             torch.from_numpy(g["z_predictor"]).cuda())
        torch.cuda.synchronize()
        np.testing.assert_allclose(mean_x.reshape(B, K, 9).cpu().numpy(), g["pred"], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(xs.reshape(B, K, n, 9).cpu().numpy(), g["proc"], rtol=1e-3, atol=5e-3)


@pytest.mark.parametrize("B,K", [(1, 1), (3, 7), (5, 50)])
def test_score_and_divergence_vs_autograd(nets, B, K):
    """gp_score_div: score and eps^T (d score / d x) eps of cond_ode_likelihood (samplers.py:49-62) against torch autograd on
    the oracle's network; ragged tiles, rows of one tile spanning several clouds."""
    snet, _ = nets
    sd = go.make_state_dict(0, "score")
    gen = torch.Generator().manual_seed(21)
    pf = torch.randn(B, 1024, generator=gen).abs()
    pose = torch.randn(B * K, 9, generator=gen)
    eps = torch.randn(B * K, 9, generator=gen) * 3.0
    for t in (1e-5, 0.2, 0.9):
        ref_s, ref_d = go.score_and_divergence(sd, pf.repeat_interleave(K, 0), pose, torch.ones(B * K, 1) * t, eps)
        cvec = snet.cloud_embed(pf.cuda())
        tvec = snet.time_embed(torch.tensor([t], device="cuda"))
        sigma = torch.tensor([0.01 * 5000.0 ** t], device="cuda")
        s, d = snet.score_and_divergence(cvec, K, pose.cuda(), eps.cuda(), tvec[0], sigma)
        np.testing.assert_allclose(s.cpu().numpy(), ref_s.numpy(), rtol=NET_RTOL, atol=NET_RTOL * float(ref_s.abs().max()))
        np.testing.assert_allclose(d.cpu().numpy(), ref_d.numpy(), rtol=NET_RTOL, atol=NET_RTOL * float(ref_d.abs().max()))
        # same score as the plain evaluation
        s0 = snet.evaluate(cvec, K, pose.cuda(), tvec[0], sigma, "score")
        np.testing.assert_allclose(s.cpu().numpy(), s0.cpu().numpy(), rtol=1e-6, atol=1e-6 * float(ref_s.abs().max()))


def test_energy_model_score_golden(nets, golden):
    """Score of the energy model (energynet.py:200-222, autograd in the reference) against the imported reference (fixture G13)
    and against the oracle's autograd on larger, ragged shapes."""
    _, enet = nets
    g = golden("g13_energy_score.npz")
    pf, pose = torch.from_numpy(g["pts_feat"]).cuda(), torch.from_numpy(g["pose"]).cuda()
    for i, t in enumerate(g["t"]):
        cvec = enet.cloud_embed(pf)
        tvec = enet.time_embed(torch.tensor([float(t)], device="cuda"))
        sigma = torch.tensor([0.01 * 5000.0 ** float(t)], device="cuda")
        s = enet.energy_score(cvec, 1, pose, tvec[0], sigma).cpu().numpy()
        np.testing.assert_allclose(s, g[f"score_{i}"], rtol=NET_RTOL, atol=NET_RTOL * np.abs(g[f"score_{i}"]).max())
    sd = go.make_state_dict(0, "energy")
    gen = torch.Generator().manual_seed(31)
    for B, K in [(3, 7), (5, 50)]:
        pfc = torch.randn(B, 1024, generator=gen).abs()
        x = torch.randn(B * K, 9, generator=gen)
        t = 0.3
        ref_s, ref_e = go.energy_score(sd, pfc.repeat_interleave(K, 0), x, torch.ones(B * K, 1) * t)
        cvec = enet.cloud_embed(pfc.cuda())
        tvec = enet.time_embed(torch.tensor([t], device="cuda"))
        sigma = torch.tensor([0.01 * 5000.0 ** t], device="cuda")
        s, e = enet.energy_score(cvec, K, x.cuda(), tvec[0], sigma, with_energy=True)
        np.testing.assert_allclose(s.cpu().numpy(), ref_s.numpy(), rtol=NET_RTOL, atol=NET_RTOL * float(ref_s.abs().max()))
        np.testing.assert_allclose(e.cpu().numpy(), ref_e.numpy(), rtol=NET_RTOL, atol=NET_RTOL * float(ref_e.abs().max()))


def test_energy_agent_score_mode():
    """GFObjectPose.forward(mode='score') on the energy agent (posenet.py:154-157) returns that gradient."""
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    ea = PoseNet(get_config(posenet_mode="energy"))
    sd = go.make_state_dict(0, "energy")
    ea.load_state_dict(sd)
    gen = torch.Generator().manual_seed(2)
    pf, pose = torch.randn(4, 1024, generator=gen).abs(), torch.randn(4, 9, generator=gen)
    data = {"pts_feat": pf.cuda(), "sampled_pose": pose.cuda(), "t": torch.ones(4, 1, device="cuda") * 0.2}
    got = ea.net(data, mode="score").cpu().numpy()
    ref, _ = go.energy_score(sd, pf, pose, torch.ones(4, 1) * 0.2)
    np.testing.assert_allclose(got, ref.numpy(), rtol=NET_RTOL, atol=NET_RTOL * float(ref.abs().max()))
