"""GPU parity of the 32-row score tile (8 waves) - chosen by the tile rule for 4096 < rows <= 8192 and 12288 < rows:
score / energy evaluation, the grouped PC sampler at the bench shape (2 x 64 clouds x 50 candidates) and the ODE sampler,
all against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go

NET_RTOL = 2e-4


@pytest.fixture(scope="module")
def nets():
    from genpose_amd.scorenet import ScoreNetHIP
    return ScoreNetHIP(go.make_state_dict(0, "score"), "cuda"), ScoreNetHIP(go.make_state_dict(0, "energy"), "cuda")


@pytest.mark.parametrize("B,K", [(100, 50), (128, 50), (97, 53)])  # 5000 / 6400 / 5141 rows (ragged last tile, 3 clouds per tile)
def test_score_and_energy_rows(nets, B, K):
    from genpose_amd import _lib
    assert _lib.lib().gp_score_tile_rows(B * K) == 32
    snet, enet = nets
    gen = torch.Generator().manual_seed(11)
    pf = torch.randn(B, 1024, generator=gen).abs()
    pose = torch.randn(B * K, 9, generator=gen)
    for net, sd, mode, fwd in ((snet, go.make_state_dict(0, "score"), "score", go.score_forward),
                               (enet, go.make_state_dict(0, "energy"), "energy", go.energy_forward)):
        for t in (1e-5, 0.4):
            ref = fwd(sd, pf.repeat_interleave(K, 0), pose, torch.ones(B * K, 1) * t).numpy()
            cvec = net.cloud_embed(pf.cuda())
            tvec = net.time_embed(torch.tensor([t], device="cuda"))
            sigma = torch.tensor([0.01 * 5000.0 ** t], device="cuda")
            got = net.evaluate(cvec, K, pose.cuda(), tvec[0], sigma, mode).cpu().numpy()
            np.testing.assert_allclose(got, ref, rtol=NET_RTOL, atol=NET_RTOL * np.abs(ref).max())


def test_grouped_pc_sampler_bench_shape(nets):
    """Two batches of 64 clouds x 50 candidates in one launch chain (6400 rows, 32-row tiles): each batch equals the oracle's
    PC sampler run on that batch alone (its own batch-mean gradient norm)."""
    from genpose_amd.samplers import PCSampler
    snet, _ = nets
    sd = go.make_state_dict(0, "score")
    B1, K, n, G = 64, 50, 6, 2
    R1 = B1 * K
    gen = torch.Generator().manual_seed(5)
    pf = torch.randn(G * B1, 1024, generator=gen).abs()
    centre = torch.randn(G * B1, 3, generator=gen) * 0.3
    init_x = torch.randn(G * R1, 9, generator=gen) * 50.0
    init_x[R1:] *= 0.2  # the two batches see very different gradient norms
    z1, z2 = torch.randn(n, G * R1, 9, generator=gen), torch.randn(n, G * R1, 9, generator=gen)
    smp = PCSampler(snet, G * B1, K, n, "cuda", use_graph=True, groups=G)
    assert smp.tile == 32
    cvec = snet.cloud_embed(pf.cuda())
    for _ in range(2):
        _, mean_x = smp.run(cvec, centre.cuda(), init_x.cuda(), z1.cuda(), z2.cuda())
        torch.cuda.synchronize()
        got = mean_x.cpu()
        for g in range(G):
            rows = slice(g * R1, (g + 1) * R1)
            feat_rows = pf[g * B1:(g + 1) * B1].repeat_interleave(K, 0)
            _, ref = go.pc_sampler(lambda x, t: go.score_forward(sd, feat_rows, x, t), init_x[rows], centre[g * B1:(g + 1) * B1].repeat_interleave(K, 0),
                                   n, z1[:, rows], z2[:, rows])
            np.testing.assert_allclose(got[rows].numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()), err_msg=f"batch {g}")


def test_ode_sampler_32_row_tiles(nets):
    from genpose_amd.samplers import ODESampler
    snet, _ = nets
    sd = go.make_state_dict(0, "score")
    B, K, T0 = 90, 50, 0.3   # 4500 rows
    from genpose_amd import _lib
    assert _lib.lib().gp_score_tile_rows(B * K) == 32
    gen = torch.Generator().manual_seed(9)
    pf = torch.randn(B, 1024, generator=gen).abs()
    centre = torch.randn(B, 3, generator=gen) * 0.3
    init_x = torch.randn(B * K, 9, generator=gen) * float(go.ve_sigma(torch.tensor(T0)))
    log = []
    feat_rows = pf.repeat_interleave(K, 0)
    _, ref, nfev = go.ode_sampler(lambda x, t: go.score_forward(sd, feat_rows, x, t), init_x, centre.repeat_interleave(K, 0), T0, log=log)
    smp = ODESampler(snet, B, K, "cuda")
    cvec = snet.cloud_embed(pf.cuda())
    _, x = smp.run(cvec, centre.cuda(), init_x.cuda(), T0)
    torch.cuda.synchronize()
    assert int(smp.last_stats["nfev"]) == nfev
    scale = max(1.0, float(ref.abs().max()))
    np.testing.assert_allclose(x.cpu().numpy(), ref.numpy(), rtol=0, atol=5e-4 * scale)


@pytest.mark.parametrize("G,B1,K", [(2, 4, 8), (3, 2, 16), (2, 64, 50)])
def test_grouped_ode_sampler_keeps_per_batch_step_control(nets, G, B1, K):
    """G batches share every launch of the RK45 driver (gp_rk45_phase_grouped) but each keeps its own controller: every batch's
    accept / reject sequence, evaluation count and result are what it gets when solved alone (the last case runs on 32-row
    tiles, the stand-alone solves on 16-row tiles: same schedule, results within the ODE tolerance)."""
    from genpose_amd.samplers import ODESampler
    snet, _ = nets
    T0 = 0.3
    gen = torch.Generator().manual_seed(100 + G)
    pf = torch.randn(G * B1, 1024, generator=gen).abs()
    centre = torch.randn(G * B1, 3, generator=gen) * 0.3
    R1 = B1 * K
    init_x = torch.randn(G * R1, 9, generator=gen) * float(go.ve_sigma(torch.tensor(T0)))
    for g in range(G):
        init_x[g * R1:(g + 1) * R1] *= 1.0 + 0.7 * g  # different scales -> different step sequences per batch
    cvec = snet.cloud_embed(pf.cuda())
    grouped = ODESampler(snet, G * B1, K, "cuda", groups=G)
    _, xg = grouped.run(cvec, centre.cuda(), init_x.cuda(), T0)
    torch.cuda.synchronize()
    alone = ODESampler(snet, B1, K, "cuda")
    accs = []
    for g in range(G):
        rows, cl = slice(g * R1, (g + 1) * R1), slice(g * B1, (g + 1) * B1)
        _, xa = alone.run(cvec[cl], centre[cl].cuda(), init_x[rows].cuda(), T0)
        torch.cuda.synchronize()
        sa, sg = alone.last_stats, grouped.group_stats[g]
        accs.append(tuple(int(v) for v in sa["log_acc"]))
        assert int(sg["status"]) == 1 and int(sg["nfev"]) == int(sa["nfev"]), (g, sg["nfev"], sa["nfev"])
        assert [int(v) for v in sg["log_acc"]] == [int(v) for v in sa["log_acc"]]
        np.testing.assert_allclose(sg["log_t"], sa["log_t"], rtol=1e-4, atol=1e-9)  # tile size changes the summation order of the error norm
        np.testing.assert_allclose(sg["log_err"], sa["log_err"], rtol=2e-2, atol=1e-4)
        scale = max(1.0, float(xa.abs().max()))
        np.testing.assert_allclose(xg[rows].cpu().numpy(), xa.cpu().numpy(), rtol=0, atol=5e-4 * scale)
    assert len(set(accs)) > 1 or G == 1  # the batches really took different step sequences


# ------------------------------------------------------------------------------------------------------------------------------------
# 64-row tiles (round 4): the plan for launches that are 1-2 partly filled rounds of 32-row tiles (12 800 rows = 400 tiles = 1.56 rounds
# of the 256 CUs -> 200 tiles = one round).  Forced here (tile = 64) at sizes the oracle finishes quickly.
@pytest.mark.parametrize("B,K", [(100, 50), (97, 53), (40, 3), (3, 200)])  # 5000 / 5141 rows (ragged last tile); tiles spanning 22 clouds (operands from global memory); one cloud over many tiles
def test_score_and_energy_rows_64_row_tiles(nets, B, K):
    snet, enet = nets
    gen = torch.Generator().manual_seed(13)
    pf = torch.randn(B, 1024, generator=gen).abs()
    pose = torch.randn(B * K, 9, generator=gen)
    for net, sd, mode, fwd in ((snet, go.make_state_dict(0, "score"), "score", go.score_forward),
                               (enet, go.make_state_dict(0, "energy"), "energy", go.energy_forward)):
        t = 0.4
        ref = fwd(sd, pf.repeat_interleave(K, 0), pose, torch.ones(B * K, 1) * t).numpy()
        cvec = net.cloud_embed(pf.cuda())
        tvec = net.time_embed(torch.tensor([t], device="cuda"))
        sigma = torch.tensor([0.01 * 5000.0 ** t], device="cuda")
        got = net.evaluate(cvec, K, pose.cuda(), tvec[0], sigma, mode, tile=64)
        base = net.evaluate(cvec, K, pose.cuda(), tvec[0], sigma, mode, tile=16)
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=NET_RTOL, atol=NET_RTOL * np.abs(ref).max())
        np.testing.assert_allclose(got.cpu().numpy(), base.cpu().numpy(), rtol=0, atol=2e-6 * np.abs(ref).max())  # same pre-activations, sums in another order


def test_samplers_on_64_row_tiles(nets):
    """PC: two batches of 64 clouds x 50 candidates per launch on 64-row tiles, each against the oracle's run of that batch alone.
    ODE: the 64-row stages take the 16-row solve's schedule, evaluation for evaluation."""
    from genpose_amd.samplers import ODESampler, PCSampler
    snet, _ = nets
    sd = go.make_state_dict(0, "score")
    B1, K, n, G = 64, 50, 6, 2
    R1 = B1 * K
    gen = torch.Generator().manual_seed(15)
    pf = torch.randn(G * B1, 1024, generator=gen).abs()
    centre = torch.randn(G * B1, 3, generator=gen) * 0.3
    init_x = torch.randn(G * R1, 9, generator=gen) * 50.0
    init_x[R1:] *= 0.2
    z1, z2 = torch.randn(n, G * R1, 9, generator=gen), torch.randn(n, G * R1, 9, generator=gen)
    smp = PCSampler(snet, G * B1, K, n, "cuda", groups=G, tile=64)
    assert smp.tile == 64 and smp.kernel_name == "pc_step_kernel<64>"
    cvec = snet.cloud_embed(pf.cuda())
    for _ in range(2):
        _, mean_x = smp.run(cvec, centre.cuda(), init_x.cuda(), z1.cuda(), z2.cuda())
    torch.cuda.synchronize()
    got = mean_x.cpu()
    for g in range(G):
        rows = slice(g * R1, (g + 1) * R1)
        feat_rows = pf[g * B1:(g + 1) * B1].repeat_interleave(K, 0)
        _, ref = go.pc_sampler(lambda x, t: go.score_forward(sd, feat_rows, x, t), init_x[rows], centre[g * B1:(g + 1) * B1].repeat_interleave(K, 0), n,
                               z1[:, rows], z2[:, rows])
        np.testing.assert_allclose(got[rows].numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()), err_msg=f"batch {g}")
    T0 = 0.3
    y0 = torch.randn(G * R1, 9, generator=gen) * float(go.ve_sigma(torch.tensor(T0)))
    res = {}
    for tile in (16, 64):
        ode = ODESampler(snet, G * B1, K, "cuda", groups=G, tile=tile)
        _, x = ode.run(cvec, centre.cuda(), y0.cuda(), T0)
        res[tile] = (x.cpu().numpy(), [int(s_["nfev"]) for s_ in ode.group_stats])
    assert res[64][1] == res[16][1], (res[64][1], res[16][1])
    scale = max(1.0, float(np.abs(res[16][0]).max()))
    np.testing.assert_allclose(res[64][0], res[16][0], rtol=0, atol=5e-4 * scale)
