"""The small-size core parity tests on OTHER networks than the seed-0 one every other test uses: weight seeds 1 and 2, and a
checkpoint-shaped stress of seed 0 (BatchNorm running variances spread over four decades, a fifth of the channels dead after their ReLU,
output layers at a trained scale).  HIP path against the CPU oracle holding the same state dict, the tolerances of the seed-0 tests."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go

from test_gpu_sampler import ODE_ROT_ATOL, ODE_RTOL


def _stress(sd, seed=11):
    g = torch.Generator().manual_seed(seed)
    out = {k: v.clone() for k, v in sd.items()}
    for k in list(out):
        if k.endswith("bn.bn.running_var"):
            out[k] = out[k] * torch.pow(10.0, torch.rand(out[k].shape, generator=g) * 4 - 2)     # 1e-2 .. 1e2 of the seeded value
        elif k.endswith("bn.bn.bias") and "layer2" not in k:
            dead = torch.rand(out[k].shape, generator=g) < 0.2
            out[k] = torch.where(dead, torch.full_like(out[k], -50.0), out[k])                     # channel is zero after the ReLU
        elif k.endswith(".2.weight") and "fusion_tail" in k:
            out[k] = out[k] * 4.0
    return out


def _weights(which, mode):
    if which == "stress":
        return _stress(go.make_state_dict(0, mode), 11 if mode == "score" else 12)
    return go.make_state_dict(int(which), mode)


def _agent(sd, mode, sampler, steps=None):
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    a = PoseNet(get_config(posenet_mode=mode, sampler_mode=[sampler], sampling_steps=steps))
    a.load_state_dict(sd)
    return a


@pytest.mark.parametrize("which", ["1", "2", "stress"])
def test_core_path_on_other_networks(which):
    from genpose_amd import reward, synth
    sd, sde = _weights(which, "score"), _weights(which, "energy")
    B, K, n = 6, 10, 20
    pts_np = synth.make_batch(B, start=8800)
    pts_cpu, pts = torch.from_numpy(pts_np), torch.from_numpy(pts_np).cuda()
    cen = pts_cpu.mean(dim=1)
    gen = torch.Generator().manual_seed(5)
    prior = torch.randn(B * K, 9, generator=gen)
    z1, z2 = torch.randn(n, B * K, 9, generator=gen), torch.randn(n, B * K, 9, generator=gen)
    # encoder
    pc = _agent(sd, "score", "pc", n)
    pc.net.prior_fn = lambda shape, T=1.0: prior * (0.01 * 5000.0 ** T)
    data = {"pts": pts, "pts_center": pts.mean(dim=1)}
    got_pc = pc.pred_func(data, K, save_path=None, noise=(z1.cuda(), z2.cuda()))
    ref_feat = go.encoder_forward(sd, pts_cpu).numpy()
    np.testing.assert_allclose(data["pts_feat"].cpu().numpy(), ref_feat, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(ref_feat).max())))
    # score network at four times, on the oracle's features
    psn = pc.net.pose_score_net
    feat_r = torch.from_numpy(ref_feat).repeat_interleave(K, 0)
    for t in (1e-5, 0.15, 0.55, 1.0):
        x = prior * 0.3
        ref = go.score_forward(sd, feat_r, x, torch.ones(B * K, 1) * t).numpy()
        got = pc.net({"pts_feat": feat_r.cuda(), "sampled_pose": x.cuda(), "t": torch.ones(B * K, 1).cuda() * t}, mode="score").cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max(), err_msg=f"score at t = {t}")
    # PC sampler, 20 steps, injected draws
    ref_pc, _, _ = go.pred_func(sd, pts_cpu, cen, K, "pc", prior, sampling_steps=n, z_langevin=z1, z_predictor=z2)
    np.testing.assert_allclose(got_pc.cpu().numpy(), ref_pc.numpy(), rtol=1e-3, atol=1e-3 * max(1.0, float(ref_pc.abs().max())))
    # ODE sampler from T0 = 0.55
    ode = _agent(sd, "score", "ode")
    ode.net.prior_fn = pc.net.prior_fn
    got_ode = ode.pred_func({"pts": pts, "pts_center": pts.mean(dim=1)}, K, save_path=None, T0=0.55)
    ref_ode, _, nfev = go.pred_func(sd, pts_cpu, cen, K, "ode", prior, T0=0.55)
    assert abs(int(ode.net.last_sampler.last_stats["nfev"]) - nfev) <= 6
    g, r = got_ode.cpu().numpy(), ref_ode.numpy()
    np.testing.assert_allclose(g[..., :6], r[..., :6], rtol=0, atol=ODE_ROT_ATOL)
    np.testing.assert_allclose(g[..., 6:], r[..., 6:], rtol=0, atol=ODE_RTOL * max(1.0, float(np.abs(r[..., 6:]).max())))
    # energies of the device's candidates, exact ranking, aggregation
    ea = _agent(sde, "energy", "ode")
    energy = ea.get_energy({"pts": pts, "pts_center": pts.mean(dim=1)}, got_ode, T=1e-5)
    ref_e = go.get_energy(sde, pts_cpu, cen, got_ode.cpu(), T=1e-5).numpy()
    np.testing.assert_allclose(energy.cpu().numpy(), ref_e, rtol=5e-4, atol=5e-4 * np.abs(ref_e).max())
    rk = reward.rank_aggregate(got_ode, energy, ratio=0.6)
    e_cpu = energy.cpu()
    for c in range(2):
        assert torch.equal(rk["order"][:, :, c].cpu().long(), torch.sort(e_cpu[:, :, c], dim=1, descending=True, stable=True).indices)
    _, qt = go.aggregate_sorted(go.pose9_to_RT(rk["sorted_poses"].cpu()), ratio=0.6)
    a = rk["avg_pose"].cpu().numpy()
    np.testing.assert_allclose(a[:, 4:], qt.numpy()[:, 4:], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(qt.numpy()[:, 4:]).max())))
    assert np.all(np.abs(np.sum(a[:, :4] * qt.numpy()[:, :4], axis=1)) > 1 - 1e-5)
