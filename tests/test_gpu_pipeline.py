"""GPU: the two-stream pipelined predictor returns, for every batch, exactly what the sequential agent returns."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go


def test_pipelined_equals_sequential():
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.pipeline import PipelinedPCPredictor
    from genpose_amd.posenet_agent import PoseNet
    B, K, n, NB = 4, 6, 12, 5
    agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=n))
    agent.load_state_dict(go.make_state_dict(0, "score"))
    gen = torch.Generator().manual_seed(0)
    batches = [torch.from_numpy(synth.make_batch(B, start=10 * i)).cuda() for i in range(NB)]
    priors = [torch.randn(B * K, 9, generator=gen) for _ in range(NB)]
    noises = [(torch.randn(n, B * K, 9, generator=gen).cuda(), torch.randn(n, B * K, 9, generator=gen).cuda()) for _ in range(NB)]
    seq = []
    for i in range(NB):
        agent.net.prior_fn = lambda shape, T=1.0, i=i: priors[i] * 50.0
        seq.append(agent.pred_func({"pts": batches[i], "pts_center": batches[i].mean(dim=1)}, K, save_path=None, noise=noises[i]).clone())
    pipe = PipelinedPCPredictor(agent, B, K, n)
    for _ in range(2):  # second pass reuses slots / replays the captured graph
        got = pipe.run(batches, prior_noise=[p.cuda() for p in priors], noise=noises)
        torch.cuda.synchronize()
        for i in range(NB):
            assert torch.equal(got[i], seq[i]), f"batch {i}"
    # oracle spot check on the first batch
    ref, _, _ = go.pred_func(go.make_state_dict(0, "score"), batches[0].cpu(), batches[0].cpu().mean(dim=1), K, "pc", priors[0],
                             sampling_steps=n, z_langevin=noises[0][0].cpu(), z_predictor=noises[0][1].cpu())
    np.testing.assert_allclose(got[0].cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("G", [2, 3])
def test_batches_sharing_a_launch_keep_their_own_coupling(G):
    """G batches per encoder pass / sampler launch chain (gp_pc_step_grouped): every batch's result is what it gets alone -
    the batch-mean gradient norm of the Langevin corrector is per batch, not per launch."""
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.pipeline import PipelinedPCPredictor
    from genpose_amd.posenet_agent import PoseNet
    B, K, n, NB = 4, 8, 10, 5  # 32 rows per batch; NB = 5 leaves a ragged tail for both G
    agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=n))
    agent.load_state_dict(go.make_state_dict(0, "score"))
    gen = torch.Generator().manual_seed(1)
    batches = [torch.from_numpy(synth.make_batch(B, start=7 * i)).cuda() for i in range(NB)]
    # very different prior scales per batch: a launch-wide mean would visibly change every batch's step size
    priors = [torch.randn(B * K, 9, generator=gen) * (1.0 + 3.0 * i) for i in range(NB)]
    noises = [(torch.randn(n, B * K, 9, generator=gen).cuda(), torch.randn(n, B * K, 9, generator=gen).cuda()) for _ in range(NB)]
    alone = PipelinedPCPredictor(agent, B, K, n).run(batches, prior_noise=[p.cuda() for p in priors], noise=noises)
    torch.cuda.synchronize()
    alone = [a.clone() for a in alone]
    pipe = PipelinedPCPredictor(agent, B, K, n, batches_per_launch=G)
    for _ in range(2):
        got = pipe.run(batches, prior_noise=[p.cuda() for p in priors], noise=noises)
        torch.cuda.synchronize()
        for i in range(NB):
            scale = float(alone[i].abs().max())
            np.testing.assert_allclose(got[i].cpu().numpy(), alone[i].cpu().numpy(), rtol=1e-5, atol=1e-5 * scale, err_msg=f"batch {i}")


def test_grouped_tile_rule():
    from genpose_amd import _lib
    l = _lib.lib()
    assert l.gp_pc_tile_rows(1, 64, 50) == 16      # 3200 rows: one 16-row tile per CU
    assert l.gp_pc_tile_rows(2, 64, 50) == 32      # 6400 rows: 200 32-row tiles
    assert l.gp_pc_tile_rows(2, 3, 10) == 16 or l.gp_pc_tile_rows(2, 3, 10) < 0  # 30 rows per batch: no tile divides it
    assert l.gp_pc_tile_rows(2, 3, 10) < 0
    assert l.gp_pc_tile_rows(2, 8, 10) == 16       # 80 rows per batch
