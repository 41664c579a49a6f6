"""GPU parity: sampling FROM the energy model (GFObjectPose.sample on a posenet_mode='energy' agent) and get_energy(T=None).

The reference's samplers call `score_model(data)`; for the energy network that is the autograd gradient of the inner-product energy
(energynet.py:200-222), not f/sigma.  The oracle restates exactly that (`go.energy_score`), so a sampler that silently used f/sigma
fails these tests by orders of magnitude."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go


def make_agent(sampler, steps=None):
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    agent = PoseNet(get_config(posenet_mode="energy", sampler_mode=[sampler], sampling_steps=steps))
    agent.load_state_dict(go.make_state_dict(0, "energy"))
    return agent


def _setup(B, K, start):
    from genpose_amd import synth
    pts = torch.from_numpy(synth.make_batch(B, start=start))
    sde = go.make_state_dict(0, "energy")
    feat_r = go.encoder_forward(sde, pts).repeat_interleave(K, 0)
    cen_r = pts.mean(dim=1).repeat_interleave(K, 0)
    score_fn = lambda x, t: go.energy_score(sde, feat_r, x, t)[0]
    return pts, sde, score_fn, cen_r


def test_pc_sampling_from_the_energy_model():
    B, K, n = 3, 6, 12
    pts, sde, score_fn, cen_r = _setup(B, K, 300)
    gen = torch.Generator().manual_seed(8)
    prior = torch.randn(B * K, 9, generator=gen)
    z1, z2 = torch.randn(n, B * K, 9, generator=gen), torch.randn(n, B * K, 9, generator=gen)
    ref_xs, ref = go.pc_sampler(score_fn, prior * 50.0, cen_r, n, z1, z2)
    agent = make_agent("pc", n)
    agent.net.prior_fn = lambda shape, T=1.0: prior * 50.0
    data = {"pts": pts.cuda(), "pts_center": pts.cuda().mean(dim=1)}
    pred, proc = agent.pred_func(data, K, save_path=None, return_process=True, noise=(z1.cuda(), z2.cuda()))
    assert pred.dtype == torch.float32 and pred.shape == (B, K, 9) and proc.shape == (B, K, n, 9)
    # rotation block (unit columns): absolute; translation block: relative to its scale (the random-weight energy model drives
    # translations to 1e5 within a dozen steps)
    got, want = pred.reshape(-1, 9).cpu().numpy(), ref.numpy()
    np.testing.assert_allclose(got[:, :6], want[:, :6], rtol=0, atol=2e-3)
    np.testing.assert_allclose(got[:, 6:], want[:, 6:], rtol=0, atol=1e-3 * max(1.0, float(np.abs(want[:, 6:]).max())))
    gp, wp = proc.reshape(B * K, n, 9).cpu().numpy(), ref_xs.numpy()
    np.testing.assert_allclose(gp[..., :6], wp[..., :6], rtol=0, atol=2e-3)
    np.testing.assert_allclose(gp[..., 6:], wp[..., 6:], rtol=0, atol=1e-3 * max(1.0, float(np.abs(wp[..., 6:]).max())))
    # the score agent's kernels (f/sigma) would give something else entirely: guard against the silent wrong path
    feat_r = go.encoder_forward(sde, pts).repeat_interleave(K, 0)
    f_over_sigma = lambda x, t: go._trunk(sde, feat_r, x, t) / go.ve_sigma(t)
    _, other = go.pc_sampler(f_over_sigma, prior * 50.0, cen_r, n, z1, z2)
    assert float((other - ref).abs().max()) > 100 * 1e-3 * float(ref.abs().max())
    # string dispatch of the reference: net(data, mode='pc_sample')
    rows = {"pts_feat": data["pts_feat"], "pts_center": data["pts_center"], "_repeat": K}
    agent.net.prior_fn = lambda shape, T=1.0: prior * 50.0
    xs2, res2 = agent.net(rows, mode="pc_sample")
    assert res2.shape == (B * K, 9) and xs2.shape == (B * K, n, 9)


def test_ode_sampling_from_the_energy_model():
    B, K, T0 = 2, 5, 0.3
    pts, sde, score_fn, cen_r = _setup(B, K, 340)
    gen = torch.Generator().manual_seed(9)
    prior = torch.randn(B * K, 9, generator=gen) * float(go.ve_sigma(torch.tensor(T0)))
    _, ref, nfev = go.ode_sampler(score_fn, prior, cen_r, T0)
    agent = make_agent("ode")
    agent.net.prior_fn = lambda shape, T=1.0: prior
    pred = agent.pred_func({"pts": pts.cuda(), "pts_center": pts.cuda().mean(dim=1)}, K, save_path=None, T0=T0)
    assert pred.dtype == torch.float64 and pred.shape == (B, K, 9)
    got = pred.reshape(-1, 9).cpu().numpy()
    np.testing.assert_allclose(got[:, :6], ref.numpy()[:, :6], rtol=0, atol=2e-3)
    np.testing.assert_allclose(got[:, 6:], ref.numpy()[:, 6:], rtol=0, atol=5e-4 * max(1.0, float(ref[:, 6:].abs().max())))
    assert abs(agent.net.last_energy_ode_stats["nfev"] - nfev) <= 0.15 * nfev


def test_get_energy_with_random_T_per_cloud():
    """posenet_agent.py:504-509: T=None draws one T in {1e-5..9e-5} per cloud on the CPU generator."""
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    B, K = 12, 7
    agent = PoseNet(get_config(posenet_mode="energy"))
    sde = go.make_state_dict(0, "energy")
    agent.load_state_dict(sde)
    pts = torch.from_numpy(synth.make_batch(B, start=60))
    gen = torch.Generator().manual_seed(2)
    poses = torch.randn(B, K, 9, generator=gen)
    torch.manual_seed(1234)
    energy = agent.get_energy({"pts": pts.cuda(), "pts_center": pts.cuda().mean(dim=1)}, poses.cuda(), T=None)
    T = agent.last_T_samples.cpu()
    torch.manual_seed(1234)
    # the division by 1e5 happens on the device (as in the reference, .type_as(pts_feat) comes first): torch's scalar division
    # there is a multiplication by the reciprocal, one ulp off the host's quotient
    assert torch.allclose(T, torch.randint(1, 10, (B, 1)).float() / 1e5, rtol=1e-6, atol=0) and len(torch.unique(T)) > 1
    feat_r = go.encoder_forward(sde, pts).repeat_interleave(K, 0)
    p = poses.reshape(B * K, 9).clone()
    p[:, -3:] -= pts.mean(dim=1).repeat_interleave(K, 0)
    ref = go.energy_forward(sde, feat_r, p, T.repeat_interleave(K, 0)).reshape(B, K, 2).numpy()
    np.testing.assert_allclose(energy.cpu().numpy(), ref, rtol=5e-4, atol=5e-4 * np.abs(ref).max())


def test_per_row_times_are_refused_not_truncated():
    from genpose_amd import synth
    agent = make_agent("pc", 4)
    pts = torch.from_numpy(synth.make_batch(2, start=5)).cuda()
    data = {"pts": pts, "pts_center": pts.mean(dim=1)}
    feat = agent.net(data, mode="pts_feature")
    t = torch.tensor([[1e-5]] * 3 + [[2e-5]] * 3, device="cuda")
    with pytest.raises(NotImplementedError):
        agent.net({"pts_feat": feat, "sampled_pose": torch.zeros(6, 9, device="cuda"), "t": t, "_repeat": 3}, mode="energy")
    with pytest.raises(NotImplementedError):
        agent.net({"pts_feat": feat, "sampled_pose": torch.zeros(6, 9, device="cuda"), "t": t, "_repeat": 3}, mode="score")


def test_energy_model_samplers_are_device_resident_and_grouped():
    """The energy model's samplers are the score model's device-resident ones (one captured launch chain / the RK45 driver) with the
    forward + vector-Jacobian pass inside the kernels: replays are bit-identical, and two batches sharing the launches keep their
    own batch-global couplings (each equals the oracle's sampler run on it alone)."""
    import genpose_amd
    from genpose_amd.samplers import ODESampler, PCSampler
    from genpose_amd.scorenet import ScoreNetHIP
    import os
    assert not os.path.exists(os.path.join(os.path.dirname(genpose_amd.__file__), "energy_sampling.py"))  # the host-driven loops are gone
    sde = go.make_state_dict(0, "energy")
    net = ScoreNetHIP(sde, "cuda")
    G, B1, K, n = 2, 2, 8, 6
    R1 = B1 * K
    gen = torch.Generator().manual_seed(21)
    pf = torch.randn(G * B1, 1024, generator=gen).abs()
    centre = torch.randn(G * B1, 3, generator=gen) * 0.3
    init_x = torch.randn(G * R1, 9, generator=gen) * 50.0
    init_x[R1:] *= 0.3
    z1, z2 = torch.randn(n, G * R1, 9, generator=gen), torch.randn(n, G * R1, 9, generator=gen)
    cvec = net.cloud_embed(pf.cuda())
    smp = PCSampler(net, G * B1, K, n, "cuda", use_graph=True, groups=G, model="energy")
    assert smp.tile == 16 and smp.kernel_name == "pc_step_kernel<16,energy>"
    outs = []
    for _ in range(2):
        _, mean_x = smp.run(cvec, centre.cuda(), init_x.cuda(), z1.cuda(), z2.cuda())
        torch.cuda.synchronize()
        outs.append(mean_x.clone())
    assert torch.equal(outs[0], outs[1]) and smp.graph is not None
    for g in range(G):
        rows = slice(g * R1, (g + 1) * R1)
        feat_rows = pf[g * B1:(g + 1) * B1].repeat_interleave(K, 0)
        _, ref = go.pc_sampler(lambda x, t: go.energy_score(sde, feat_rows, x, t)[0], init_x[rows], centre[g * B1:(g + 1) * B1].repeat_interleave(K, 0),
                               n, z1[:, rows], z2[:, rows])
        got, want = outs[0][rows].cpu().numpy(), ref.numpy()
        np.testing.assert_allclose(got[:, :6], want[:, :6], rtol=0, atol=2e-3, err_msg=f"batch {g}")
        np.testing.assert_allclose(got[:, 6:], want[:, 6:], rtol=0, atol=1e-3 * max(1.0, float(np.abs(want[:, 6:]).max())), err_msg=f"batch {g}")
    with pytest.raises(ValueError):
        PCSampler(net, 640, 50, n, "cuda", groups=10, model="energy", tile=32)  # the backward pass has no 32-row tile form (16, or the chain form)
    # ODE: two batches, each with its own step controller == the oracle's solve of that batch alone (attempt for attempt)
    T0 = 0.3
    y0 = torch.randn(G * R1, 9, generator=gen) * float(go.ve_sigma(torch.tensor(T0)))
    ode = ODESampler(net, G * B1, K, "cuda", groups=G, model="energy")
    _, x = ode.run(cvec, centre.cuda(), y0.cuda(), T0)
    for g in range(G):
        rows = slice(g * R1, (g + 1) * R1)
        feat_rows = pf[g * B1:(g + 1) * B1].repeat_interleave(K, 0)
        _, ref, nfev = go.ode_sampler(lambda xx, t: go.energy_score(sde, feat_rows, xx, t)[0], y0[rows], centre[g * B1:(g + 1) * B1].repeat_interleave(K, 0), T0)
        got = x[rows].cpu().numpy()
        np.testing.assert_allclose(got[:, :6], ref.numpy()[:, :6], rtol=0, atol=2e-3, err_msg=f"batch {g}")
        np.testing.assert_allclose(got[:, 6:], ref.numpy()[:, 6:], rtol=0, atol=5e-4 * max(1.0, float(ref[:, 6:].abs().max())), err_msg=f"batch {g}")
        assert abs(int(ode.group_stats[g]["nfev"]) - nfev) <= max(12, 0.1 * nfev), (g, ode.group_stats[g]["nfev"], nfev)


def test_likelihood_rows_per_cloud_and_solver_reuse():
    """cond_ode_likelihood with several candidates per cloud (the reference repeats the cloud features row by row): the same rows
    with per-cloud embeddings addressed by row / K and with one embedding per row give the same solve - the error norm runs over the
    same state vector - and a reused solver (captured attempts) reproduces its own result bit for bit.  (Agreement with the imported
    reference's solve: fixture G12, tests/test_gpu_sampler.py.)"""
    from genpose_amd.likelihood import cond_ode_likelihood
    from genpose_amd.samplers import ODESampler
    from genpose_amd.scorenet import ScoreNetHIP
    net = ScoreNetHIP(go.make_state_dict(0, "score"), "cuda")
    B, K = 2, 3
    gen = torch.Generator().manual_seed(31)
    pf = torch.randn(B, 1024, generator=gen).abs().cuda()
    x = torch.randn(B * K, 9, generator=gen).cuda()
    probe = (torch.randn(B * K, 9, generator=gen) * 50.0).cuda()
    cvec = net.cloud_embed(pf)
    solver = ODESampler(net, B, K, "cuda", model="likelihood")
    s1, s2, s3 = {}, {}, {}
    z1, l1 = cond_ode_likelihood(net, cvec, K, x, probe, rtol=1e-4, atol=1e-4, stats=s1, solver=solver)
    z2, l2 = cond_ode_likelihood(net, cvec, K, x, probe, rtol=1e-4, atol=1e-4, stats=s2, solver=solver)
    z3, l3 = cond_ode_likelihood(net, cvec.repeat_interleave(K, 0).contiguous(), 1, x, probe, rtol=1e-4, atol=1e-4, stats=s3)
    assert z1.dtype == torch.float64 and l1.shape == (B * K,) and torch.isfinite(l1).all()
    assert torch.equal(l1, l2) and torch.equal(z1, z2) and s1 == s2
    assert s1["nfev"] == s3["nfev"] and s1["nfev"] == 2 + 6 * s1["attempts"]
    np.testing.assert_allclose(l1.cpu().numpy(), l3.cpu().numpy(), rtol=1e-9, atol=1e-9 * float(l1.abs().max()))
    with pytest.raises(RuntimeError):
        solver.run(cvec, torch.zeros(B, 3, device="cuda"), x, 0.5)  # a likelihood solver does not sample
