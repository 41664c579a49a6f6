"""GPU parity at the sizes BASELINE.json's configs are quoted on (the sizes bench.py times), not scaled-down stand-ins:

  configs[1]  G x 64 clouds x 1024 pts (G = bench.py's default request batching), K = 50, PC sampler with 100 steps: one whole batch against
              the CPU oracle, every batch through size-independent properties, G batches per launch (32-row tiles) against one batch per
              launch (16-row tiles)
  encoder     64 and 320 clouds against the oracle
  configs[2]  256 clouds: score model + energy model + ranking + aggregation in FullPipelinePredictor against the agents called one
              after the other, the oracle on a 16-cloud slice, and the exact ranking permutation on all 256
  load_ckpt   through a reference-layout .pth file on the device

Tolerances (fp32 everywhere, random-weight networks): the PC sampler is a 100-step stochastic recursion whose noise is injected, so two
correct fp32 implementations stay within round-off amplified by the recursion - stated per test below.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go
from oracle import parallel as opar  # the oracle's encoder over hundreds of clouds: cached per cloud, slices side by side on the host cores

ENC_RTOL, ENC_ATOL = 2e-4, 2e-4
# PC-100 end state, two correct fp32 implementations against each other (HIP vs oracle, or 32-row vs 16-row tiles whose
# reductions sum in a different order).  The recursion renormalises the rotation columns every step (samplers.py:142-143), which
# amplifies round-off for the rare row whose column passes close to zero: the bulk of the 10^5 rotation components agrees to 1e-4,
# single outliers reach a few 1e-3 (measured on MI355X: max 3.7e-3, translations 8e-7 relative).  Stated tolerance:
#   rotation block (unit columns, absolute): 99.9 % of the components within 1e-3, every component within 1e-2
#   translation block: within 1e-4 of its scale
PC100_ROT_P999, PC100_ROT_MAX, PC100_TRANS_RTOL = 1e-3, 1e-2, 1e-4


def make_agent(mode, sampler="pc", steps=None):
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    agent = PoseNet(get_config(posenet_mode=mode, sampler_mode=[sampler], sampling_steps=steps))
    agent.load_state_dict(go.make_state_dict(0, mode))
    return agent


def _pose_errors(got, ref):
    """(99.9th percentile and max abs error of the rotation block, max error of the translation block relative to its scale)"""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    d = np.abs(got[..., :6] - ref[..., :6]).reshape(-1)
    trans = float(np.abs(got[..., 6:] - ref[..., 6:]).max() / max(1.0, np.abs(ref[..., 6:]).max()))
    return float(np.quantile(d, 0.999)), float(d.max()), trans


def _assert_pc100_close(got, ref, what):
    p999, mx, trans = _pose_errors(got, ref)
    print(f"{what}: rotation p99.9 {p999:.2e} max {mx:.2e}, translation (rel) {trans:.2e}")
    assert p999 < PC100_ROT_P999 and mx < PC100_ROT_MAX and trans < PC100_TRANS_RTOL, (what, p999, mx, trans)


class _host_threads:
    """torch's intra-op thread count for a CPU-oracle sampler run: the default on the GPU box is one thread per host core (256), which is
    SLOWER than a few dozen on 3 200 - 12 800-row GEMMs (measured on the box: 55 ms per 12 800-row evaluation at 16 threads, 77 at 32, 4 800 at 256)."""

    def __init__(self, n=16):
        self.n = max(1, min(n, os.cpu_count() or n))

    def __enter__(self):
        self.prev = torch.get_num_threads()
        torch.set_num_threads(self.n)

    def __exit__(self, *exc):
        torch.set_num_threads(self.prev)


def _oracle_pc(sd, pts_cpu, K, prior, n, z1, z2):
    """go.pred_func(sampler='pc') for the seed-0 score weights `sd`, with the oracle's encoder walked in slices (clouds are independent; its
    grouped tensors are 12 MB per cloud; oracle/parallel.py) -> mean_x [B,K,9]."""
    B = pts_cpu.shape[0]
    feat = torch.from_numpy(opar.encoder_features("score", pts_cpu))
    feat_r = feat.repeat_interleave(K, 0)
    cen_r = pts_cpu.mean(dim=1).repeat_interleave(K, 0)
    with _host_threads():
        _, x = go.pc_sampler(lambda x, t: go.score_forward(sd, feat_r, x, t), prior * float(go.ve_sigma(1.0)), cen_r, n, z1, z2)
    return x.reshape(B, K, 9)


def _check_pose_properties(pred):
    """Size-independent properties of a PC-sampler result [.., 9]: finite, the two rotation columns unit-norm and orthogonal
    (normalize_rotation is the last thing applied to mean_x, samplers.py:157-158)."""
    assert torch.isfinite(pred).all()
    a, b = pred[..., :3].double(), pred[..., 3:6].double()
    assert float((a.norm(dim=-1) - 1).abs().max()) < 1e-5
    assert float((b.norm(dim=-1) - 1).abs().max()) < 1e-5
    assert float((a * b).sum(-1).abs().max()) < 1e-5


def test_config1_as_timed():
    """BASELINE configs[1] exactly as bench.py times it: PipelinedPCPredictor(batches_per_launch=G) on G x 64 synthetic clouds (G = the
    bench's default: 10 -> 32 000 rows per launch, 32-row tiles), K = 50, 100 PC steps, with injected prior / Langevin / predictor draws."""
    import bench
    from genpose_amd import synth
    from genpose_amd.pipeline import PipelinedPCPredictor
    B1, K, n, G = 64, 50, 100, bench.DEFAULT_BATCHES_PER_LAUNCH  # the bench's own default request batching
    R1 = B1 * K
    agent = make_agent("score", "pc", n)
    batches = [torch.from_numpy(synth.make_batch(B1, start=B1 * i)).cuda() for i in range(G)]
    gen = torch.Generator().manual_seed(2024)
    priors = [torch.randn(R1, 9, generator=gen) for _ in range(G)]
    noises = [(torch.randn(n, R1, 9, generator=gen), torch.randn(n, R1, 9, generator=gen)) for _ in range(G)]
    noises_dev = [(a.cuda(), b.cuda()) for a, b in noises]
    pipe5 = PipelinedPCPredictor(agent, B1, K, n, batches_per_launch=G)
    # 32 000 rows per launch: the chain form of the trunk (128-row workgroups, csrc/trunk_chain.h), or 32-row tiles
    assert pipe5._sampler(0, G).tile in (32, 128) and pipe5._sampler(0, G).R == G * R1
    got5 = [g.clone() for g in pipe5.run(batches, prior_noise=[p.cuda() for p in priors], noise=noises_dev)]
    torch.cuda.synchronize()
    for g in got5:
        assert g.shape == (B1, K, 9) and g.dtype == torch.float32
        _check_pose_properties(g)
    # a second replay of the captured graph gives the same bits (deterministic reductions)
    again = pipe5.run(batches, prior_noise=[p.cuda() for p in priors], noise=noises_dev)
    torch.cuda.synchronize()
    for a, b in zip(again, got5):
        assert torch.equal(a, b)
    # request batching does not change a batch's result: one batch per launch (16-row tiles, own launch chain) agrees
    pipe1 = PipelinedPCPredictor(agent, B1, K, n, batches_per_launch=1)
    assert pipe1._sampler(0, 1).tile == 16
    got1 = pipe1.run(batches, prior_noise=[p.cuda() for p in priors], noise=noises_dev)
    torch.cuda.synchronize()
    _assert_pc100_close(torch.stack(got5).cpu().numpy(), torch.stack(list(got1)).cpu().numpy(), f"configs[1] {G}-per-launch vs 1-per-launch")
    # one WHOLE batch (the middle one) against the CPU oracle: encoder + 100-step PC sampler with the same draws
    i = G // 2
    pts_cpu = batches[i].cpu()
    ref = _oracle_pc(go.make_state_dict(0, "score"), pts_cpu, K, priors[i], n, noises[i][0], noises[i][1])
    _assert_pc100_close(got5[i].cpu().numpy(), ref.numpy(), f"configs[1] batch {i} (3200 rows x 100 steps) vs oracle")


def test_ode_100_as_timed():
    """The secondary mode of the bench line (`ode_100`: cond_ode_sampler(sampling_steps=100), T0 = 0.55) exactly as bench.py times it:
    GroupedODEPredictor with the bench's request batching (10 x 64 clouds = 32 000 rows per launch: the RK45 stage kernels run in the
    chain form of the trunk, every batch with its own step controller), injected prior draws.  Every batch: finite, unit and orthogonal
    rotation columns; a second solve gives the same bits; and request batching does not change a batch's solve - against one batch per
    launch (16-row tile form) the same evaluation count (one attempt of slack), poses within the ODE tolerance of the small tests (rotation block 2e-3
    absolute, translations 5e-4 of their scale)."""
    import bench
    from genpose_amd import synth
    from genpose_amd.pipeline import GroupedODEPredictor
    B1, K, G, T0 = 64, 50, bench.DEFAULT_BATCHES_PER_LAUNCH, 0.55
    agent = make_agent("score", "ode", 100)
    batches = [torch.from_numpy(synth.make_batch(B1, start=7000 + B1 * i)).cuda() for i in range(G)]
    gen = torch.Generator().manual_seed(77)
    priors = [torch.randn(B1 * K, 9, generator=gen).cuda() for _ in range(G)]
    pg = GroupedODEPredictor(agent, B1, K, T0=T0, batches_per_launch=G)
    got = [g.clone() for g in pg.run(batches, prior_noise=priors)]
    nfev = list(pg.last_nfev)
    assert pg._sampler(G).tile in (32, 128) and len(nfev) == G and all(150 < v < 400 for v in nfev), nfev
    for g in got:
        assert g.shape == (B1, K, 9) and g.dtype == torch.float64
        _check_pose_properties(g.float())
    again = pg.run(batches, prior_noise=priors)
    for a, b in zip(again, got):
        assert torch.equal(a, b)
    p1 = GroupedODEPredictor(agent, B1, K, T0=T0, batches_per_launch=1)
    one = [g.clone() for g in p1.run(batches, prior_noise=priors)]
    # same controller decisions; an error norm within round-off of 1.0 may flip ONE accept / reject between the two forms of the trunk
    # (2e-7 apart): at most one attempt (6 evaluations) of difference per batch is tolerated, the poses are held to the tolerance either way
    assert p1._sampler(1).tile == 16 and all(abs(a_ - b_) <= 6 for a_, b_ in zip(p1.last_nfev, nfev)), (p1.last_nfev, nfev)
    a, b = torch.stack(got).cpu().numpy(), torch.stack(one).cpu().numpy()
    np.testing.assert_allclose(a[..., :6], b[..., :6], rtol=0, atol=2e-3, err_msg="rotation block, 10 per launch vs 1 per launch")
    np.testing.assert_allclose(a[..., 6:], b[..., 6:], rtol=0, atol=5e-4 * np.abs(b[..., 6:]).max(), err_msg="translations")


@pytest.mark.parametrize("B", [64, 320, 448])
def test_encoder_vs_oracle_at_bench_sizes(B):
    """The encoder at the batch sizes bench.py runs it at (64 = one batch, 320 = five batches per launch) and at 448 clouds.  Clouds are
    independent, so the oracle walks the batch in slices (its grouped tensors are 12 MB per cloud).  The GroupAll level takes whole
    rounds of 256 clouds on its ring kernel (one cloud per workgroup) and the rest on 32-row tiles: 64 = tiles only, 320 = 256 + 64,
    448 = two rounds of the ring kernel, the second three quarters full."""
    from genpose_amd import synth
    from genpose_amd.encoder import Pointnet2EncoderHIP
    sd = go.make_state_dict(0, "score")
    enc = Pointnet2EncoderHIP(sd, "cuda")
    pts = synth.make_batch(B, start=7000)
    got = enc.forward(torch.from_numpy(pts).cuda()).cpu().numpy()
    assert got.shape == (B, 1024) and np.isfinite(got).all()
    ref = opar.encoder_features("score", pts)  # every one of the B clouds (the three sizes share their first clouds: computed once)
    step = 32
    for s in range(0, B, step):
        np.testing.assert_allclose(got[s:s + step], ref[s:s + step], rtol=ENC_RTOL, atol=ENC_ATOL, err_msg=f"clouds {s}..{s + step}")


def test_config2_full_pipeline_256():
    """BASELINE configs[2]: 256 clouds through score model -> energy model -> ranking -> top-60 % aggregation."""
    from genpose_amd import reward, synth
    from genpose_amd.pipeline import FullPipelinePredictor
    B, K, n = 256, 50, 100
    sa, ea = make_agent("score", "pc", n), make_agent("energy")
    pts = torch.from_numpy(synth.make_batch(B, start=9000)).cuda()
    centre = pts.mean(dim=1)
    gen = torch.Generator().manual_seed(77)
    prior = torch.randn(B * K, 9, generator=gen)
    z1, z2 = torch.randn(n, B * K, 9, generator=gen).cuda(), torch.randn(n, B * K, 9, generator=gen).cuda()
    fp = FullPipelinePredictor(sa, ea, B, K, n)
    out = fp.run(pts, prior_noise=prior.cuda(), noise=(z1, z2))
    torch.cuda.synchronize()
    pred, energy = out["pred_pose"], out["energy"]
    assert pred.shape == (B, K, 9) and energy.shape == (B, K, 2) and out["avg_pose"].shape == (B, 7)
    _check_pose_properties(pred)
    assert torch.isfinite(energy).all() and torch.isfinite(out["avg_pose"]).all()
    # (1) == the agents called one after the other (the reference's order of calls)
    sa.net.prior_fn = lambda shape, T=1.0: prior * 50.0
    seq_pred = sa.pred_func({"pts": pts, "pts_center": centre}, K, save_path=None, noise=(z1, z2))
    seq_energy = ea.get_energy({"pts": pts, "pts_center": centre}, seq_pred, T=1e-5)
    seq = reward.rank_aggregate(seq_pred, seq_energy, ratio=0.6)
    torch.cuda.synchronize()
    assert torch.equal(seq_pred, pred)
    assert torch.equal(seq_energy, energy)
    assert torch.equal(seq["order"], out["order"]) and torch.equal(seq["avg_pose"], out["avg_pose"])
    # (2) ranking is the exact (stable, descending) permutation on all 256 clouds, and the sorted tensors are consistent with it
    e_cpu, order = energy.cpu(), out["order"].cpu().long()
    for c in range(2):
        ref = torch.sort(e_cpu[:, :, c], dim=1, descending=True, stable=True)
        assert torch.equal(order[:, :, c], ref.indices)
        assert torch.equal(out["sorted_energy"][:, :, c].cpu(), ref.values)
    bi = torch.arange(B).unsqueeze(1).expand(B, K)
    p_cpu = pred.cpu()
    want = p_cpu[bi, order[:, :, 0]].clone()
    want[:, :, 6:] = p_cpu[bi, order[:, :, 1]][:, :, 6:]
    assert torch.equal(out["sorted_poses"].cpu(), want)
    # (3) the oracle on a 16-cloud slice: energies of the device's own candidates, aggregation of the device's ranking
    sl = slice(100, 116)
    ref_e = go.get_energy(go.make_state_dict(0, "energy"), pts[sl].cpu(), centre[sl].cpu(), p_cpu[sl], T=1e-5).numpy()
    np.testing.assert_allclose(e_cpu[sl].numpy(), ref_e, rtol=5e-4, atol=5e-4 * np.abs(ref_e).max())
    _, qt = go.aggregate_sorted(go.pose9_to_RT(out["sorted_poses"][sl].cpu()), ratio=0.6)
    avg = out["avg_pose"][sl].cpu().numpy()
    np.testing.assert_allclose(avg[:, 4:], qt.numpy()[:, 4:], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(qt.numpy()[:, 4:]).max())))
    assert np.all(np.abs(np.sum(avg[:, :4] * qt.numpy()[:, :4], axis=1)) > 1 - 1e-5)
    # (4) the score model's sampler at this size against the oracle (all 12 800 rows are coupled through the batch-mean score norm)
    ref_pred = _oracle_pc(go.make_state_dict(0, "score"), pts.cpu(), K, prior, n, z1.cpu(), z2.cpu())
    _assert_pc100_close(p_cpu.numpy(), ref_pred.numpy(), "configs[2] sampler (12800 rows x 100 steps) vs oracle")


def test_load_ckpt_through_a_pth_on_the_device(tmp_path):
    """PoseNet.load_ckpt (posenet_agent.py:143-173) with a reference-layout checkpoint file: the agent it fills gives the same
    bits as one filled from the state dict in memory."""
    from genpose_amd import synth
    sd = go.make_state_dict(0, "score")
    path = os.path.join(str(tmp_path), "ckpt_genpose.pth")
    torch.save({"model_state_dict": {("module." + k if i % 2 else k): v for i, (k, v) in enumerate(sd.items())},
                "optimizer_state_dict": {}, "scheduler_state_dict": {}, "clock": {"epoch": 3}}, path)
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    a = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=8))
    a.load_ckpt(model_dir=path, model_path=True, load_model_only=True)
    b = make_agent("score", "pc", 8)
    with pytest.raises(ValueError):
        a.load_ckpt(model_dir=os.path.join(str(tmp_path), "missing.pth"), model_path=True)
    pts = torch.from_numpy(synth.make_batch(3, start=21)).cuda()
    gen = torch.Generator().manual_seed(3)
    prior = torch.randn(3 * 6, 9, generator=gen)
    z = (torch.randn(8, 18, 9, generator=gen).cuda(), torch.randn(8, 18, 9, generator=gen).cuda())
    outs = []
    for ag in (a, b):
        ag.net.prior_fn = lambda shape, T=1.0: prior * 50.0
        outs.append(ag.pred_func({"pts": pts, "pts_center": pts.mean(dim=1)}, 6, save_path=None, noise=z).clone())
    assert torch.equal(outs[0], outs[1])
    ref, _, _ = go.pred_func(sd, pts.cpu(), pts.cpu().mean(dim=1), 6, "pc", prior, sampling_steps=8, z_langevin=z[0].cpu(), z_predictor=z[1].cpu())
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()))


def test_full_pipeline_request_batching_keeps_every_batch():
    """FullPipelinePredictor.run_many: G batches share every launch (encoders, sampler chain with per-batch coupling, energy evaluation,
    ranking); every batch gets what run() gives it alone - up to the tile size of the sampler (32-row tiles for the group, 16-row for one
    batch: the norm partials sum in a different order)."""
    from genpose_amd import synth
    from genpose_amd.pipeline import FullPipelinePredictor
    B, K, n, G, NB = 64, 50, 20, 2, 3  # NB = 3 leaves a ragged tail (one batch alone in the last group)
    sa, ea = make_agent("score", "pc", n), make_agent("energy")
    batches = [torch.from_numpy(synth.make_batch(B, start=300 * i)).cuda() for i in range(NB)]
    gen = torch.Generator().manual_seed(5)
    priors = [torch.randn(B * K, 9, generator=gen).cuda() * (1.0 + i) for i in range(NB)]  # different scales: a launch-wide mean would show
    noises = [(torch.randn(n, B * K, 9, generator=gen).cuda(), torch.randn(n, B * K, 9, generator=gen).cuda()) for _ in range(NB)]
    fp = FullPipelinePredictor(sa, ea, B, K, n, batches_per_launch=G)
    many = fp.run_many(batches, prior_noise=priors, noise=noises)
    many = [{k: v.clone() for k, v in m.items()} for m in many]
    assert len(many) == NB
    for i in range(NB):
        one = fp.run(batches[i], prior_noise=priors[i], noise=noises[i])
        torch.cuda.synchronize()
        assert many[i]["pred_pose"].shape == (B, K, 9) and many[i]["avg_pose"].shape == (B, 7)
        scale = float(one["pred_pose"].abs().max())
        np.testing.assert_allclose(many[i]["pred_pose"].cpu().numpy(), one["pred_pose"].cpu().numpy(), rtol=0, atol=1e-3 * scale, err_msg=f"batch {i}")
        e1 = one["energy"].cpu().numpy()
        np.testing.assert_allclose(many[i]["energy"].cpu().numpy(), e1, rtol=0, atol=2e-3 * np.abs(e1).max(), err_msg=f"batch {i}")
        if i == NB - 1:  # the ragged tail runs alone, on the same 16-row tiles as run(): identical bits
            assert torch.equal(many[i]["pred_pose"], one["pred_pose"]) and torch.equal(many[i]["order"], one["order"])


def test_config2_run_many_as_timed():
    """BASELINE configs[2] exactly as bench.py's `full_pipeline_256` leg times it: FullPipelinePredictor.run_many with FIVE batches of 256
    clouds per launch (64 000 rows, 2000 32-row tiles), K = 50, 100 PC steps, distinct clouds in every batch, injected draws.  Every batch
    must be what run() gives it alone; the ranking must be the exact stable permutation on all 1 280 clouds; one batch goes against the
    CPU oracle (sampler over its 12 800 coupled rows; energies and aggregation on a slice)."""
    from genpose_amd import synth
    from genpose_amd.pipeline import FullPipelinePredictor
    B, K, n, G = 256, 50, 100, 5
    R = B * K
    sa, ea = make_agent("score", "pc", n), make_agent("energy")
    batches = [torch.from_numpy(synth.make_batch(B, start=20000 + B * i)).cuda() for i in range(G)]
    gen = torch.Generator().manual_seed(404)
    priors = [torch.randn(R, 9, generator=gen) for _ in range(G)]
    noises = [(torch.randn(n, R, 9, generator=gen), torch.randn(n, R, 9, generator=gen)) for _ in range(G)]
    priors_dev = [p.cuda() for p in priors]
    noises_dev = [(a.cuda(), b.cuda()) for a, b in noises]
    fp = FullPipelinePredictor(sa, ea, B, K, n, batches_per_launch=G)
    assert fp._sampler(G).tile in (32, 128) and fp._sampler(G).R == G * R  # 64 000 rows: the chain form (128-row workgroups)
    many = [{k: v.clone() for k, v in m.items()} for m in fp.run_many(batches, prior_noise=priors_dev, noise=noises_dev)]
    torch.cuda.synchronize()
    assert len(many) == G
    for i, m in enumerate(many):
        assert m["pred_pose"].shape == (B, K, 9) and m["energy"].shape == (B, K, 2) and m["avg_pose"].shape == (B, 7)
        _check_pose_properties(m["pred_pose"])
        assert torch.isfinite(m["energy"]).all() and torch.isfinite(m["avg_pose"]).all()
        # exact (stable, descending) permutation of THIS batch's energies, on every one of its 256 clouds
        e_cpu, order = m["energy"].cpu(), m["order"].cpu().long()
        for c in range(2):
            ref = torch.sort(e_cpu[:, :, c], dim=1, descending=True, stable=True)
            assert torch.equal(order[:, :, c], ref.indices), (i, c)
            assert torch.equal(m["sorted_energy"][:, :, c].cpu(), ref.values)
        bi = torch.arange(B).unsqueeze(1).expand(B, K)
        p_cpu = m["pred_pose"].cpu()
        want = p_cpu[bi, order[:, :, 0]].clone()
        want[:, :, 6:] = p_cpu[bi, order[:, :, 1]][:, :, 6:]
        assert torch.equal(m["sorted_poses"].cpu(), want)
        # == the batch alone (32-row tiles, its own launch chain): the coupling stays per batch
        one = fp.run(batches[i], prior_noise=priors_dev[i], noise=noises_dev[i])
        torch.cuda.synchronize()
        _assert_pc100_close(p_cpu.numpy(), one["pred_pose"].cpu().numpy(), f"configs[2] run_many batch {i} vs run() alone")
        e1 = one["energy"].cpu().numpy()
        np.testing.assert_allclose(e_cpu.numpy(), e1, rtol=0, atol=2e-3 * np.abs(e1).max(), err_msg=f"batch {i}")
    # a second replay gives the same bits
    again = fp.run_many(batches, prior_noise=priors_dev, noise=noises_dev)
    torch.cuda.synchronize()
    for a, b in zip(again, many):
        assert torch.equal(a["pred_pose"], b["pred_pose"]) and torch.equal(a["order"], b["order"]) and torch.equal(a["avg_pose"], b["avg_pose"])
    # batch 3 against the oracle
    i = 3
    m = many[i]
    pts_cpu = batches[i].cpu()
    ref_pred = _oracle_pc(go.make_state_dict(0, "score"), pts_cpu, K, priors[i], n, noises[i][0], noises[i][1])
    _assert_pc100_close(m["pred_pose"].cpu().numpy(), ref_pred.numpy(), f"configs[2] run_many batch {i} (12800 of 64000 rows x 100 steps) vs oracle")
    sl = slice(40, 56)
    cen = pts_cpu.mean(dim=1)
    ref_e = go.get_energy(go.make_state_dict(0, "energy"), pts_cpu[sl], cen[sl], m["pred_pose"][sl].cpu(), T=1e-5).numpy()
    np.testing.assert_allclose(m["energy"][sl].cpu().numpy(), ref_e, rtol=5e-4, atol=5e-4 * np.abs(ref_e).max())
    _, qt = go.aggregate_sorted(go.pose9_to_RT(m["sorted_poses"][sl].cpu()), ratio=0.6)
    avg = m["avg_pose"][sl].cpu().numpy()
    np.testing.assert_allclose(avg[:, 4:], qt.numpy()[:, 4:], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(qt.numpy()[:, 4:]).max())))
    assert np.all(np.abs(np.sum(avg[:, :4] * qt.numpy()[:, :4], axis=1)) > 1 - 1e-5)


def test_drop_in_eval_single_as_timed():
    """The reference's own evaluation call sequence at its own batch shape, through the AGENT API only, exactly as bench.py's
    `drop_in_eval_single` leg times it (scripts/eval_single.sh + evaluation_single.py:356-489): 256 clouds per batch, K = 50,
    pred_func with the ODE sampler from T0 = 0.55 (adaptive RK45 over all 12 800 coupled rows) -> get_energy(T = 1e-5) -> ranking ->
    top-60 % aggregation.  Three calls: launch by launch, the call that captures the encoder passes, a replay - identical bits; then
    against the CPU oracle: the solver's evaluation count and the poses of ALL 12 800 rows (the batch-global error norm couples
    them), energies and aggregation on a 16-cloud slice, encoder features on a 32-cloud slice."""
    from genpose_amd import reward, synth
    from genpose_amd.runner import make_batch_sample
    B, K, T0 = 256, 50, 0.55
    sa, ea = make_agent("score", "ode", None), make_agent("energy")
    pts = torch.from_numpy(synth.make_batch(B, start=20000)).cuda()
    prior = torch.randn(B * K, 9, generator=torch.Generator().manual_seed(11))
    sa.net.prior_fn = lambda shape, T=1.0: prior * (0.01 * 5000.0 ** T)
    runs = []
    for _ in range(3):
        data = make_batch_sample(pts)
        pred = sa.pred_func(data=data, repeat_num=K, save_path=None, T0=T0)
        energy = ea.get_energy(data=data, pose_samples=pred, T=1e-5)
        r = reward.rank_aggregate(pred, energy, ratio=0.6)
        runs.append((pred.clone(), energy.clone(), r["order"].clone(), r["avg_pose"].clone(), data["pts_feat"].clone(),
                     int(sa.net.last_sampler.last_stats["nfev"])))
    torch.cuda.synchronize()
    assert all(ent.get("graph") is not None for ent in sa.net.pts_encoder._pass_graphs.values())  # the passes are graph replays by now
    assert all(ent.get("graph") is not None for ent in ea.net.pts_encoder._pass_graphs.values())
    for later in runs[1:]:
        for a, b in zip(runs[0][:5], later[:5]):
            assert torch.equal(a, b)
        assert later[5] == runs[0][5]
    pred, energy, order, avg, feat, nfev = runs[0]
    assert pred.dtype == torch.float64 and pred.shape == (B, K, 9) and energy.shape == (B, K, 2)
    _check_pose_properties(pred.float())
    # ---- oracle
    sd, sde = go.make_state_dict(0, "score"), go.make_state_dict(0, "energy")
    pts_cpu = pts.cpu()
    ref_feat = torch.from_numpy(opar.encoder_features("score", pts_cpu))
    np.testing.assert_allclose(feat.cpu().numpy(), ref_feat.numpy(), rtol=ENC_RTOL, atol=ENC_ATOL)
    feat_r = ref_feat.repeat_interleave(K, 0)
    cen = pts_cpu.mean(dim=1)
    with _host_threads():
        _, ref_x, ref_nfev = go.ode_sampler(lambda x, t: go.score_forward(sd, feat_r, x, t), prior * go.ve_sigma(T0),
                                            cen.repeat_interleave(K, 0), T0)
    assert abs(nfev - ref_nfev) <= 6, (nfev, ref_nfev)  # at most one attempt of difference (an error norm within round-off of 1)
    got, ref = pred.cpu().numpy().reshape(B * K, 9), ref_x.numpy()
    np.testing.assert_allclose(got[:, :6], ref[:, :6], rtol=0, atol=2e-3, err_msg="rotation block, 12800 rows")
    np.testing.assert_allclose(got[:, 6:], ref[:, 6:], rtol=0, atol=5e-4 * np.abs(ref[:, 6:]).max(), err_msg="translations")
    sl = slice(64, 80)
    ref_e = go.get_energy(sde, pts_cpu[sl], cen[sl], pred[sl].cpu(), T=1e-5).numpy()
    np.testing.assert_allclose(energy[sl].cpu().numpy(), ref_e, rtol=5e-4, atol=5e-4 * np.abs(ref_e).max())
    e_cpu = energy.cpu()
    for c in range(2):
        assert torch.equal(order[:, :, c].cpu().long(), torch.sort(e_cpu[:, :, c], dim=1, descending=True, stable=True).indices)
    sorted_poses = reward.rank_aggregate(pred, energy, ratio=0.6)["sorted_poses"]
    _, qt = go.aggregate_sorted(go.pose9_to_RT(sorted_poses[sl].cpu()), ratio=0.6)
    a = avg[sl].cpu().numpy()
    np.testing.assert_allclose(a[:, 4:], qt.numpy()[:, 4:], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(qt.numpy()[:, 4:]).max())))
    assert np.all(np.abs(np.sum(a[:, :4] * qt.numpy()[:, :4], axis=1)) > 1 - 1e-5)


@pytest.mark.parametrize("sampler,steps,T0", [("pc", 20, None), ("ode", None, 0.55)])
def test_config0_single_object(sampler, steps, T0):
    """BASELINE configs[0] - the reference's own CPU-runnable case: ONE cloud of 1024 points, 10 candidates, 20 SDE steps (PC) or the
    default adaptive ODE solve, through the agent (evaluation_single.py path) - against the oracle end to end: encoder, sampler, energies,
    ranking, aggregation.  Second and third calls (the captured encoder pass, its replay) give the first call's bits."""
    from genpose_amd import reward, synth
    B, K = 1, 10
    sa, ea = make_agent("score", sampler, steps), make_agent("energy")
    pts_np = synth.make_batch(B, start=4242)
    pts = torch.from_numpy(pts_np).cuda()
    gen = torch.Generator().manual_seed(8)
    prior = torch.randn(B * K, 9, generator=gen)
    noise_cpu = (torch.randn(steps, B * K, 9, generator=gen), torch.randn(steps, B * K, 9, generator=gen)) if sampler == "pc" else None
    noise = None if noise_cpu is None else tuple(z.cuda() for z in noise_cpu)
    sa.net.prior_fn = lambda shape, T=1.0: prior * (0.01 * 5000.0 ** T)
    outs = []
    for _ in range(3):
        data = {"pts": pts, "pts_center": pts.mean(dim=1)}
        pred = sa.pred_func(data, repeat_num=K, save_path=None, T0=T0, noise=noise)
        energy = ea.get_energy(data=data, pose_samples=pred, T=1e-5)
        r = reward.rank_aggregate(pred, energy, ratio=0.6)
        outs.append((pred.clone(), energy.clone(), r["order"].clone(), r["avg_pose"].clone()))
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))
    pred, energy, order, avg = outs[0]
    sd, sde = go.make_state_dict(0, "score"), go.make_state_dict(0, "energy")
    pts_cpu = torch.from_numpy(pts_np)
    cen = pts_cpu.mean(dim=1)
    if sampler == "pc":
        ref, _, _ = go.pred_func(sd, pts_cpu, cen, K, "pc", prior, sampling_steps=steps, z_langevin=noise_cpu[0], z_predictor=noise_cpu[1])
        np.testing.assert_allclose(pred.cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-3)
    else:
        ref, _, nfev = go.pred_func(sd, pts_cpu, cen, K, "ode", prior, T0=T0)
        assert abs(int(sa.net.last_sampler.last_stats["nfev"]) - nfev) <= 6
        got = pred.cpu().numpy()
        np.testing.assert_allclose(got[..., :6], ref.numpy()[..., :6], rtol=0, atol=2e-3)
        np.testing.assert_allclose(got[..., 6:], ref.numpy()[..., 6:], rtol=0, atol=5e-4 * max(1.0, float(ref[..., 6:].abs().max())))
    ref_e = go.get_energy(sde, pts_cpu, cen, pred.cpu(), T=1e-5).numpy()
    np.testing.assert_allclose(energy.cpu().numpy(), ref_e, rtol=5e-4, atol=5e-4 * np.abs(ref_e).max())
    e_cpu = energy.cpu()
    for c in range(2):
        assert torch.equal(order[:, :, c].cpu().long(), torch.sort(e_cpu[:, :, c], dim=1, descending=True, stable=True).indices)
    sorted_poses = reward.rank_aggregate(pred, energy, ratio=0.6)["sorted_poses"]
    _, qt = go.aggregate_sorted(go.pose9_to_RT(sorted_poses.cpu()), ratio=0.6)
    a = avg.cpu().numpy()
    np.testing.assert_allclose(a[:, 4:], qt.numpy()[:, 4:], rtol=1e-5, atol=1e-5)
    assert np.all(np.abs(np.sum(a[:, :4] * qt.numpy()[:, :4], axis=1)) > 1 - 1e-5)
