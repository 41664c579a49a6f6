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
