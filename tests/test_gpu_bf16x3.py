"""GPU: the OPT-IN split-bf16 form of the 128-196-256 grouping level (csrc/sa_bf16x3.hip, encoder precision 'bf16x3'; exploratory, round 5).
Three bf16 matrix products with fp32 accumulation per fp32 product: a . b ~= a_hi . b_hi + a_lo . b_hi + a_hi . b_lo.  Stated error bound
(the gate of the experiment): <= 3e-5 of the level's feature scale against an fp64 evaluation of the same level; the fp32 kernels sit at
~1e-6.  Centres and neighbourhoods do not depend on the precision (bit-identical)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go

GATE = 3e-5


def _level_fp64(enc, ws, B, k=2, feat_in=None):
    """Grouping level k (both scales) of the light encoder in float64 on the host, from the folded weights and the device's own level k-1
    output (or `feat_in`), centres and neighbourhoods: h1 = relu(W1f feat_j + W1x (x_j - c) + b1) -> relu(W2 h1 + b2) -> max_j relu(W3 h2 + b3)."""
    xyz = ws["new_xyz"][k - 1][:B].double().cpu()      # level k-1's centres = level k's points
    feat = (ws["feat"][k - 1][:B] if feat_in is None else feat_in[:B]).double().cpu()
    centres = ws["new_xyz"][k][:B].double().cpu()
    outs = []
    for i, sc in enumerate(enc.w.levels[k]):
        (W1, b1), (W2, b2), (W3, b3) = [(W.double(), b.double()) for W, b in sc._folded_plain]  # W1 columns: [feat..., dx, dy, dz]
        idx = ws["bq"][k][i][:B].long().cpu()      # [B,np,ns]
        bi = torch.arange(B)[:, None, None]
        nf = feat[bi, idx]                          # [B,128,ns,256]
        d = xyz[bi, idx] - centres[:, :, None, :]   # [B,128,ns,3]
        h1 = torch.relu(torch.cat([nf, d], dim=-1) @ W1.t() + b1)
        h2 = torch.relu(h1 @ W2.t() + b2)
        h3 = torch.relu(h2 @ W3.t() + b3)
        outs.append(h3.max(dim=2)[0])
    return torch.cat(outs, dim=-1)  # [B,np,2 x c3]


def _level2_fp64(enc, ws, B):
    return _level_fp64(enc, ws, B, 2)


def test_bf16x3_level_against_fp64_and_fp32():
    from genpose_amd import synth
    from genpose_amd.encoder import Pointnet2EncoderHIP
    sd = go.make_state_dict(0, "score")
    e32, ebf = Pointnet2EncoderHIP(sd, "cuda"), Pointnet2EncoderHIP(sd, "cuda", precision="bf16x3")
    pts = torch.from_numpy(synth.make_batch(6, start=1234)).cuda()
    f32, w32 = e32.forward(pts, return_intermediates=True)
    f32 = f32.clone()
    l_32 = {k: w32["feat"][k].clone() for k in (0, 1, 2)}
    fbf, wbf = ebf.forward(pts, return_intermediates=True)
    l_bf = {k: wbf["feat"][k].clone() for k in (0, 1, 2)}
    for k in range(3):  # the grouping does not depend on the precision
        assert torch.equal(w32["fps_idx"][k], wbf["fps_idx"][k]) and all(torch.equal(a, b) for a, b in zip(w32["bq"][k], wbf["bq"][k]))
    assert torch.equal(l_32[0], l_bf[0])  # nor does level 0, which stays on the fp32 pipe
    for k in (1, 2):
        # each level against fp64 ON ITS OWN INPUT (the split kernel's level 2 reads the split kernel's level 1)
        ref32, refbf = _level_fp64(e32, w32, 3, k, l_32[k - 1]), _level_fp64(e32, w32, 3, k, l_bf[k - 1])
        scale = float(ref32.abs().max())
        e_32 = float((l_32[k][:3].double().cpu() - ref32).abs().max()) / scale
        e_bf = float((l_bf[k][:3].double().cpu() - refbf).abs().max()) / scale
        print(f"level {k} vs fp64, max error / feature scale: fp32 pipe {e_32:.2e}, split bf16 {e_bf:.2e} (gate {GATE:.0e})")
        assert e_32 < 3e-6 and e_bf < GATE, (k, e_32, e_bf)
        assert not torch.equal(l_32[k], l_bf[k])  # (it IS a different arithmetic)
    # end of the encoder: the deviation stays at that level through the GroupAll level
    d = float((fbf - f32).abs().max()) / float(f32.abs().max())
    assert d < 5 * GATE, d
    # a batch size whose rows do not fill whole 32-row units of the 16-neighbourhood scale is still served (3 clouds x 128 centres x 16)
    assert torch.isfinite(ebf.forward(pts[:3].contiguous())).all()


def test_bf16x3_through_the_agent_is_opt_in():
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    sd = go.make_state_dict(0, "score")
    assert get_config().encoder_precision == "f32"
    a = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=6, encoder_precision="bf16x3"))
    a.load_state_dict(sd)
    assert a.net.pts_encoder.precision == "bf16x3"
    pts = torch.from_numpy(synth.make_batch(4, start=77)).cuda()
    pred = a.pred_func({"pts": pts, "pts_center": pts.mean(dim=1)}, repeat_num=8, save_path=None)
    assert torch.isfinite(pred).all()
    with pytest.raises(ValueError):
        from genpose_amd.encoder import Pointnet2EncoderHIP
        Pointnet2EncoderHIP(sd, "cuda", precision="fp8")


def test_split_bf16_pc_step_against_fp32_and_oracle():
    """The OPT-IN split-bf16 form of the PC sampler's score network (csrc/trunk_bf16x3.hip, PCSampler(precision='bf16x3')): the first score
    evaluation against the fp32 kernels and against an fp64 evaluation of the network (stated bound: 3e-5 of the score's scale; the fp32
    kernels: ~1e-6), a 30-step chain against the fp32 sampler and the oracle with the same draws, several batches per launch each with
    its own coupling, a ragged last workgroup, replays to the same bits."""
    from genpose_amd.samplers import PCSampler
    from genpose_amd.scorenet import ScoreNetHIP
    sd = go.make_state_dict(0, "score")
    net = ScoreNetHIP(sd, "cuda")
    n = 30
    for B, K, groups in ((5, 50, 1), (128, 50, 2), (3, 43, 1)):
        g = torch.Generator().manual_seed(B)
        feat = torch.randn(B, 1024, generator=g) * 0.5
        centre = torch.randn(B, 3, generator=g) * 0.1
        x0 = torch.randn(B * K, 9, generator=g) * 50.0
        z1, z2 = torch.randn(n, B * K, 9, generator=g), torch.randn(n, B * K, 9, generator=g)
        cvec = net.cloud_embed(feat.cuda())
        out = {}
        for prec in ("f32", "bf16x3"):
            smp = PCSampler(net, B, K, n, "cuda", groups=groups, precision=prec)
            smp.cvec.copy_(cvec), smp.centre.copy_(centre.cuda()), smp.x.copy_(x0.cuda()), smp.z1.copy_(z1.cuda()), smp.z2.copy_(z2.cuda())
            smp.launch_step(0)
            torch.cuda.synchronize()
            first = smp.score.clone()
            mean_x = smp.run(cvec, centre.cuda(), x0.cuda(), z1.cuda(), z2.cuda())[1].clone()
            again = smp.run(cvec, centre.cuda(), x0.cuda(), z1.cuda(), z2.cuda())[1]
            assert torch.equal(again, mean_x)
            out[prec] = (first, mean_x)
            assert smp.kernel_name == ("pc_step_bf16x3_kernel" if prec == "bf16x3" else smp.kernel_name)
        # first evaluation vs fp64
        sd64 = {k: v.double() for k, v in sd.items()}
        feat_r = feat.repeat_interleave(K, 0).double()
        ref = go.score_forward(sd64, feat_r, x0.double(), torch.ones(B * K, 1, dtype=torch.float64))
        scale = float(ref.abs().max())
        e32 = float((out["f32"][0].double().cpu() - ref).abs().max()) / scale
        ebf = float((out["bf16x3"][0].double().cpu() - ref).abs().max()) / scale
        print(f"B={B} K={K} groups={groups}: first score evaluation vs fp64, max error / scale: fp32 {e32:.2e}, split bf16 {ebf:.2e}")
        assert e32 < 5e-6 and ebf < GATE, (e32, ebf)
        # the chain: fp32 sampler and oracle (same draws)
        sc = float(out["f32"][1].abs().max())
        np.testing.assert_allclose(out["bf16x3"][1].cpu().numpy(), out["f32"][1].cpu().numpy(), rtol=0, atol=1e-3 * sc)
        if groups == 1:
            cen_r = centre.repeat_interleave(K, 0)
            fr = feat.repeat_interleave(K, 0)
            _, oref = go.pc_sampler(lambda x, t: go.score_forward(sd, fr, x, t), x0, cen_r, n, z1, z2)
            np.testing.assert_allclose(out["bf16x3"][1].cpu().numpy(), oref.numpy(), rtol=0, atol=1e-3 * float(oref.abs().max()))
    with pytest.raises(NotImplementedError):
        PCSampler(net, 5, 50, 4, "cuda", precision="bf16x3", model="energy")
    with pytest.raises(ValueError):
        PCSampler(net, 5, 10, 4, "cuda", precision="bf16x3")  # 10 candidates per cloud: a workgroup's 128 rows span 14 clouds


def test_sampler_precision_through_the_agent_and_the_pipeline():
    """cfg.sampler_precision = 'bf16x3' is honoured where the 128-row plan applies (>= 43 candidates per cloud, score model, PC sampler,
    uncoupled) and ignored elsewhere (the fp32 kernels run); the ODE sampler never takes it."""
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.pipeline import PipelinedPCPredictor
    from genpose_amd.posenet_agent import PoseNet
    sd = go.make_state_dict(0, "score")
    assert get_config().sampler_precision == "f32"
    pts = torch.from_numpy(synth.make_batch(200, start=31)).cuda()
    data = lambda n=200: {"pts": pts[:n].contiguous(), "pts_center": pts[:n].mean(dim=1)}
    a = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=8, sampler_precision="bf16x3"))
    a.load_state_dict(sd)
    pred = a.pred_func(data(), repeat_num=50, save_path=None)  # 10 000 rows
    assert torch.isfinite(pred).all() and a.net.last_sampler.precision == "bf16x3" and a.net.last_sampler.kernel_name == "pc_step_bf16x3_kernel"
    a.pred_func(data(), repeat_num=10, save_path=None)
    assert a.net.last_sampler.precision == "f32"  # 10 candidates per cloud: the fp32 plans
    a.pred_func(data(6), repeat_num=50, save_path=None)
    assert a.net.last_sampler.precision == "f32" and a.net.last_sampler.hsplit == 3  # 300 rows: the latency regime's fp32 plan is faster
    o = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"], sampler_precision="bf16x3"))
    o.load_state_dict(sd)
    assert torch.isfinite(o.pred_func(data(6), repeat_num=50, save_path=None, T0=0.55)).all()
    assert o.net.last_sampler.__class__.__name__ == "ODESampler"
    # request batching: groups of 64 clouds x 50 rows = 3200 = 25 x 128 rows
    b = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=8, sampler_precision="bf16x3", encoder_precision="bf16x3"))
    b.load_state_dict(sd)
    pipe = PipelinedPCPredictor(b, 64, 50, 8, batches_per_launch=2, overlap=False)
    batches = [torch.from_numpy(synth.make_batch(64, start=64 * i)).cuda() for i in range(3)]
    outs = pipe.run(batches)
    torch.cuda.synchronize()
    assert len(outs) == 3 and all(torch.isfinite(x).all() for x in outs)
    assert pipe._sampler(0, 2).precision == "f32"  # 6 400 rows per launch: below the threshold
    pipe4 = PipelinedPCPredictor(b, 64, 50, 8, batches_per_launch=4, overlap=False)  # 12 800 rows per launch
    outs4 = pipe4.run(batches + batches[:1])
    torch.cuda.synchronize()
    assert len(outs4) == 4 and pipe4._sampler(0, 4).precision == "bf16x3"
