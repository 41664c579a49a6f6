"""GPU parity of the CHAIN form of the score trunk (csrc/trunk_chain.h: one wave carries 32 rows through all layers in registers,
weights through an LDS ring, output layers on the matrix pipe) - the plan large launches take (>= ~32 000 rows) - at sizes the
oracle finishes in seconds, forced through the `tile` argument: against the CPU oracle, against the tile form, with ragged tails,
several batches per launch (per-batch coupling) and the refusals."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go

CHAIN = 128


@pytest.fixture(scope="module")
def nets():
    from genpose_amd.scorenet import ScoreNetHIP
    return ScoreNetHIP(go.make_state_dict(0, "score"), "cuda"), ScoreNetHIP(go.make_state_dict(0, "energy"), "cuda")


@pytest.mark.parametrize("B,K", [(3, 50), (41, 50), (7, 64), (130, 43)])
def test_score_and_energy_eval_chain_vs_oracle(nets, B, K):
    """R = B*K rows incl. ragged last workgroups (150, 2050, 448, 5590 rows)."""
    from genpose_amd.sde import SIGMA_MAX, SIGMA_MIN
    snet, enet = nets
    gen = torch.Generator().manual_seed(B * 100 + K)
    pf = torch.randn(B, 1024, generator=gen).abs()
    x = torch.randn(B * K, 9, generator=gen)
    for net, sd, mode, ref_fn in ((snet, go.make_state_dict(0, "score"), "score", go.score_forward),
                                  (enet, go.make_state_dict(0, "energy"), "energy", go.energy_forward)):
        for tval in (1e-5, 0.4):
            t0 = torch.full((1,), tval, device="cuda")
            cvec = net.cloud_embed(pf.cuda())
            tvec = net.time_embed(t0)[0].contiguous()
            sigma = (SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** t0).contiguous()
            got = net.evaluate(cvec, K, x.cuda(), tvec, sigma, mode, tile=CHAIN).cpu().numpy()
            tile = net.evaluate(cvec, K, x.cuda(), tvec, sigma, mode, tile=16).cpu().numpy()
            ref = ref_fn(sd, pf.repeat_interleave(K, 0), x, torch.full((B * K, 1), tval)).numpy()
            scale = np.abs(ref).max()
            np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * scale, err_msg=f"{mode} t={tval} vs oracle")
            np.testing.assert_allclose(got, tile, rtol=0, atol=2e-6 * scale, err_msg=f"{mode} t={tval} chain vs tile form")


def test_chain_refuses_what_it_cannot_stage(nets):
    """A 128-row workgroup stages cvec + tvec of at most 4 clouds: K < 43 is refused (GP_EINVAL), never mis-addressed; batches whose
    rows do not split into 128-row workgroups are refused by the sampler."""
    from genpose_amd import _lib
    from genpose_amd.samplers import PCSampler
    snet, _ = nets
    cvec = torch.zeros(40, 768, device="cuda")
    t0 = torch.full((1,), 0.5, device="cuda")
    with pytest.raises(_lib.GenposeHipError):
        snet.evaluate(cvec, 10, torch.zeros(400, 9, device="cuda"), snet.time_embed(t0)[0].contiguous(), t0, "score", tile=CHAIN)
    with pytest.raises(ValueError):
        PCSampler(snet, 2 * 3, 50, 4, "cuda", groups=2, tile=CHAIN)  # 150 rows per batch: a workgroup would straddle two batches
    assert PCSampler(snet, 3, 50, 4, "cuda", groups=1, tile=CHAIN).tile == CHAIN  # one batch: a ragged last workgroup is fine
    assert PCSampler(snet, 64, 50, 4, "cuda").tile == 16 and PCSampler(snet, 640, 50, 4, "cuda", groups=10).tile in (32, CHAIN)


@pytest.mark.parametrize("G,B1,K,n", [(2, 64, 50, 6), (1, 45, 50, 8)])
def test_pc_sampler_chain_vs_oracle(nets, G, B1, K, n):
    """The predictor-corrector chain on the chain form: every batch of a launch equals the oracle's sampler run on that batch alone
    (its own batch-mean gradient norm, reduced from one partial per WAVE); the in-process samples too; replays are bit-identical."""
    from genpose_amd.samplers import PCSampler
    snet, _ = nets
    sd = go.make_state_dict(0, "score")
    R1 = B1 * K
    gen = torch.Generator().manual_seed(11 * G + B1)
    pf = torch.randn(G * B1, 1024, generator=gen).abs()
    centre = torch.randn(G * B1, 3, generator=gen) * 0.3
    init_x = torch.randn(G * R1, 9, generator=gen) * 50.0
    if G > 1:
        init_x[R1:] *= 0.2  # the batches see very different gradient norms: a launch-wide mean would show
    z1, z2 = torch.randn(n, G * R1, 9, generator=gen), torch.randn(n, G * R1, 9, generator=gen)
    smp = PCSampler(snet, G * B1, K, n, "cuda", use_graph=True, record_traj=True, groups=G, tile=CHAIN)
    assert smp.tile == CHAIN and smp.kernel_name.startswith("pc_step_chain_kernel")
    cvec = snet.cloud_embed(pf.cuda())
    outs = []
    for _ in range(2):
        xs, mean_x = smp.run(cvec, centre.cuda(), init_x.cuda(), z1.cuda(), z2.cuda())
        torch.cuda.synchronize()
        outs.append((xs.clone(), mean_x.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    xs, got = outs[0][0].cpu(), outs[0][1].cpu()
    for g in range(G):
        rows = slice(g * R1, (g + 1) * R1)
        feat_rows = pf[g * B1:(g + 1) * B1].repeat_interleave(K, 0)
        ref_xs, ref = go.pc_sampler(lambda x, t: go.score_forward(sd, feat_rows, x, t), init_x[rows], centre[g * B1:(g + 1) * B1].repeat_interleave(K, 0),
                                    n, z1[:, rows], z2[:, rows])
        np.testing.assert_allclose(got[rows].numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()), err_msg=f"batch {g}")
        np.testing.assert_allclose(xs[rows].numpy(), ref_xs.numpy(), rtol=1e-3, atol=1e-3 * float(ref_xs.abs().max()), err_msg=f"batch {g} trajectory")
    # and the tile form on the same draws
    smp32 = PCSampler(snet, G * B1, K, n, "cuda", use_graph=False, groups=G, tile=32 if (B1 * K) % 32 == 0 else 16)
    _, m32 = smp32.run(cvec, centre.cuda(), init_x.cuda(), z1.cuda(), z2.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.numpy(), m32.cpu().numpy(), rtol=0, atol=2e-5 * float(got.abs().max()))


def test_ode_stage_kernels_chain_vs_tile_form(nets):
    """RK45 stage kernels in the chain form (rk45_stage_chain_kernel) against the tile form: two batches per launch (each with its own
    step controller), T0 = 0.55 as benched - the same accept / reject sequence, evaluation count and poses; and one ragged batch."""
    from genpose_amd.samplers import ODESampler
    snet, _ = nets
    for groups, B in ((2, 128), (1, 45)):  # 2 x 3200 rows; 2250 rows: a ragged last workgroup
        K = 50
        gen = torch.Generator().manual_seed(groups)
        cvec = torch.randn(B, 768, generator=gen).cuda()
        centre = torch.randn(B, 3, generator=gen).cuda()
        x0 = torch.randn(B * K, 9, generator=gen).cuda()
        res = {}
        for tile in (32, CHAIN):
            smp = ODESampler(snet, B, K, "cuda", groups=groups, tile=tile)
            assert smp.tile == tile
            _, x = smp.run(cvec, centre, x0, T0=0.55)
            counts = [(g["nfev"], g["n_attempts"], g["n_accepted"]) for g in smp.group_stats]
            res[tile] = (x.cpu().numpy(), counts)
        a, b = res[32], res[CHAIN]
        assert np.isfinite(b[0]).all() and len(b[1]) == groups
        # the same accept / reject sequence; an error norm within round-off of 1.0 may flip one decision between the two forms (2e-7 apart)
        assert all(abs(x[0] - y[0]) <= 6 and abs(x[1] - y[1]) <= 1 for x, y in zip(a[1], b[1])), (a[1], b[1])
        np.testing.assert_allclose(b[0][:, :6], a[0][:, :6], rtol=0, atol=2e-4)
        np.testing.assert_allclose(b[0][:, 6:], a[0][:, 6:], rtol=0, atol=2e-4 * np.abs(a[0][:, 6:]).max())
    with pytest.raises(ValueError):
        ODESampler(snet, 6, 50, "cuda", groups=2, tile=CHAIN)  # 150 rows per batch
    assert ODESampler(snet, 64, 50, "cuda").tile in (16, 32) and ODESampler(snet, 640, 50, "cuda", groups=10).tile in (32, CHAIN)

