"""GPU: the forward + vector-Jacobian right-hand sides in the CHAIN form (csrc/trunk_chain_vjp.h: the energy model's score - the gradient
of its inner-product energy, energynet.py:200-222 - inside the PC step and the RK45 stages, and the likelihood ODE's score + Hutchinson
divergence, samplers.py:22-99) against the 16-row tile form (csrc/score_bwd.h) on the same inputs and against the oracle's autograd.
The chain form is what large launches take (gp_pc_layout / gp_rk45_plan_rows); here it is forced (tile = 128) at sizes the oracle
finishes in seconds, including a ragged last workgroup and several batches per launch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go


def _net(mode):
    from genpose_amd.scorenet import ScoreNetHIP
    return ScoreNetHIP(go.make_state_dict(0, mode), "cuda")


def _close(a, b, rot_atol, trans_rtol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a[..., :6], b[..., :6], rtol=0, atol=rot_atol, err_msg=what + ": rotation block")
    np.testing.assert_allclose(a[..., 6:9], b[..., 6:9], rtol=0, atol=trans_rtol * max(1.0, float(np.abs(b[..., 6:9]).max())), err_msg=what + ": translations")


@pytest.mark.parametrize("B,K,groups", [(6, 50, 1), (4, 64, 2), (3, 128, 1)])
def test_pc_energy_model_chain_vs_tile_and_oracle(B, K, groups):
    """B x K rows: 300 rows = two full 128-row workgroups + a ragged one whose rows span clouds; two batches of 128 rows sharing the
    launches (their own batch-mean gradient norms); one cloud per workgroup."""
    from genpose_amd.samplers import PCSampler
    net, n = _net("energy"), 6
    R = B * K
    gen = torch.Generator().manual_seed(B * K)
    pf = torch.randn(B, 1024, generator=gen).abs()
    centre = torch.randn(B, 3, generator=gen) * 0.3
    x0 = torch.randn(R, 9, generator=gen) * 50.0
    z1, z2 = torch.randn(n, R, 9, generator=gen), torch.randn(n, R, 9, generator=gen)
    cvec = net.cloud_embed(pf.cuda())
    out = {}
    for tile in (16, 128):
        smp = PCSampler(net, B, K, n, "cuda", groups=groups, model="energy", tile=tile)
        assert smp.tile == tile and smp.kernel_name == ("pc_step_kernel<16,energy>" if tile == 16 else "pc_step_chain_kernel<2,energy>")
        for _ in range(2):  # launch by launch, then the captured chain
            _, m = smp.run(cvec, centre.cuda(), x0.cuda(), z1.cuda(), z2.cuda())
        torch.cuda.synchronize()
        out[tile] = (m.cpu().numpy().copy(), smp.score.cpu().numpy().copy())
    # the two forms against each other: the same network function, sums in a different order (1e-7 relative per evaluation)
    _close(out[128][0], out[16][0], 1e-4, 1e-5, "chain vs tile, final poses")
    np.testing.assert_allclose(out[128][1], out[16][1], rtol=0, atol=2e-5 * np.abs(out[16][1]).max(), err_msg="last score")
    # and against the oracle's autograd score, batch by batch
    sde = go.make_state_dict(0, "energy")
    Bg, Rg = B // groups, R // groups
    for g in range(groups):
        rows = slice(g * Rg, (g + 1) * Rg)
        feat_rows = pf[g * Bg:(g + 1) * Bg].repeat_interleave(K, 0)
        _, ref = go.pc_sampler(lambda x, t: go.energy_score(sde, feat_rows, x, t)[0], x0[rows], centre[g * Bg:(g + 1) * Bg].repeat_interleave(K, 0), n,
                               z1[:, rows], z2[:, rows])
        _close(out[128][0][rows], ref.numpy(), 2e-3, 1e-3, f"chain vs oracle, batch {g}")


def test_ode_energy_model_chain_vs_tile_and_oracle():
    from genpose_amd.samplers import ODESampler
    net = _net("energy")
    B, K, T0 = 6, 50, 0.3
    gen = torch.Generator().manual_seed(77)
    pf = torch.randn(B, 1024, generator=gen).abs()
    centre = torch.randn(B, 3, generator=gen) * 0.3
    y0 = torch.randn(B * K, 9, generator=gen) * float(go.ve_sigma(torch.tensor(T0)))
    cvec = net.cloud_embed(pf.cuda())
    res = {}
    for tile in (16, 128):
        ode = ODESampler(net, B, K, "cuda", model="energy", tile=tile)
        assert ode.tile == tile
        _, x = ode.run(cvec, centre.cuda(), y0.cuda(), T0)
        _, x2 = ode.run(cvec, centre.cuda(), y0.cuda(), T0)
        assert torch.equal(x, x2)
        res[tile] = (x.cpu().numpy(), int(ode.last_stats["nfev"]), np.array(ode.last_stats["log_err"]))
    assert abs(res[128][1] - res[16][1]) <= 6, (res[128][1], res[16][1])  # the same schedule (one attempt of slack at an error norm of 1)
    k = min(len(res[16][2]), len(res[128][2]), 10)
    np.testing.assert_allclose(res[128][2][:k], res[16][2][:k], rtol=1e-3, atol=1e-6)
    _close(res[128][0], res[16][0], 2e-4, 1e-4, "chain vs tile")
    sde = go.make_state_dict(0, "energy")
    feat_rows = pf.repeat_interleave(K, 0)
    _, ref, nfev = go.ode_sampler(lambda xx, t: go.energy_score(sde, feat_rows, xx, t)[0], y0, centre.repeat_interleave(K, 0), T0)
    _close(res[128][0], ref.numpy(), 2e-3, 5e-4, "chain vs oracle")
    assert abs(res[128][1] - nfev) <= max(12, 0.1 * nfev)


def test_likelihood_chain_vs_tile():
    """The ten-component likelihood ODE (pose + log-density, samplers.py:22-99) with the chain-form stages.  (1) The right-hand side itself
    - score and Hutchinson divergence estimate of every row, as the first two evaluations of the solve write them (f0 at t = eps, f1 after
    the trial Euler step) - against the tile form's: 1e-5 of the largest component.  (2) The whole solve (~900 attempts at rtol 1e-4 on
    a random-weight network with a probe of scale 50 - a chaotic problem: the two forms' right-hand sides differ at the 1e-7 level and the
    step-size feedback carries that into the schedule): attempt counts within 10 %, log-likelihoods within 1 % for all but a few rows."""
    from genpose_amd.likelihood import cond_ode_likelihood
    from genpose_amd.samplers import ODESampler
    net = _net("score")
    B, K = 3, 50
    R = B * K
    gen = torch.Generator().manual_seed(5)
    pf = torch.randn(B, 1024, generator=gen).abs().cuda()
    x = torch.randn(R, 9, generator=gen).cuda()
    probe = (torch.randn(R, 9, generator=gen) * 50.0).cuda()
    cvec = net.cloud_embed(pf)
    res, rhs = {}, {}
    for tile in (16, 128):
        solver = ODESampler(net, B, K, "cuda", model="likelihood", tile=tile)
        assert solver.tile == tile
        # (1) the first two right-hand-side evaluations of the solve (phases 0-2 of run_likelihood)
        solver.cvec.copy_(cvec)
        solver.centre.zero_()
        solver.probe.copy_(probe)
        y0 = solver.y.view(R, 10)
        y0[:, :9].copy_(x.double())
        y0[:, 9].zero_()
        solver._phase(0, None, t0=1e-5, t_bound=1.0, rtol=1e-4, atol=1e-4)
        solver._phase(1, None)
        solver._phase(2, None)
        torch.cuda.synchronize()
        rhs[tile] = solver.Kbuf[:2].view(2, R, 10).cpu().numpy().copy()
        st = {}
        z, ll = cond_ode_likelihood(net, cvec, K, x, probe, rtol=1e-4, atol=1e-4, stats=st, solver=solver)
        res[tile] = (z.cpu().numpy(), ll.cpu().numpy(), st)
    for q in range(2):
        for comps, what in ((slice(0, 9), "score"), (slice(9, 10), "divergence estimate")):
            a, b = rhs[128][q][:, comps], rhs[16][q][:, comps]
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-5 * np.abs(b).max(), err_msg=f"evaluation {q}, {what}")
    assert np.abs(rhs[16][0][:, 9]).max() > 0  # the divergence component is live
    assert abs(res[128][2]["nfev"] - res[16][2]["nfev"]) <= 0.1 * res[16][2]["nfev"], (res[128][2], res[16][2])
    rel = np.abs(res[128][1] - res[16][1]) / np.abs(res[16][1]).max()
    assert np.quantile(rel, 0.95) < 1e-2 and rel.max() < 1e-1, (np.quantile(rel, 0.95), rel.max())
    np.testing.assert_allclose(res[128][0], res[16][0], rtol=0, atol=2e-2 * max(1.0, np.abs(res[16][0]).max()))


def test_large_launches_take_the_chain_form():
    from genpose_amd.samplers import ODESampler, PCSampler
    net = _net("energy")
    assert PCSampler(net, 640, 50, 4, "cuda", groups=10, model="energy").tile == 128
    assert PCSampler(net, 64, 50, 4, "cuda", model="energy").tile == 16
    assert ODESampler(net, 640, 50, "cuda", groups=10, model="energy").tile == 128
    assert ODESampler(_net("score"), 640, 50, "cuda", groups=10, model="likelihood").tile == 128
    assert ODESampler(net, 64, 50, "cuda", model="energy").tile == 16
