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
