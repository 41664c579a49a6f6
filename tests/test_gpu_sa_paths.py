"""GPU: the three implementations of a grouping set-abstraction scale agree - tile kernel on the un-hoisted form
(gp_sa_mlp_max), tile kernel on the hoisted form (gp_sa_pre_mlp_max with GP_SA_NOCHAIN semantics is not selectable at run time,
so the hoisted entry point is compared as dispatched: register-chain kernels for the light config) - and match the oracle's
grouped conv+BN+ReLU+max."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go
from oracle import pn2_oracle as ops


@pytest.mark.parametrize("level,scale", [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1)])
def test_sa_scale_paths_agree(level, scale):
    from genpose_amd import _lib, synth
    from genpose_amd._lib import ptr, stream_ptr
    from genpose_amd.weights import EncoderWeights
    sd = go.make_state_dict(0, "score")
    B = 3
    pts = torch.from_numpy(synth.make_batch(B, start=70))
    _, inter = go.encoder_forward(sd, pts, return_intermediates=True)
    cfg = go.LIGHT_CFG
    # inputs of this level from the oracle's intermediates
    xyz = pts.numpy() if level == 0 else inter[level - 1]["new_xyz"]
    feats = None if level == 0 else np.ascontiguousarray(inter[level - 1]["features"].transpose(0, 2, 1))  # [B,n,C]
    new_xyz = inter[level]["new_xyz"]
    bq = inter[level][f"bq_idx{scale}"]
    n, npnt, ns = xyz.shape[1], new_xyz.shape[1], bq.shape[2]
    spec = cfg["mlps"][level][scale]
    cout_all = sum(m[-1] for m in cfg["mlps"][level])
    off = sum(m[-1] for m in cfg["mlps"][level][:scale])
    ref = inter[level]["features"][:, off:off + spec[-1], :].transpose(0, 2, 1)  # [B,np,c3]

    ew = EncoderWeights(sd, "cuda")
    sc = ew.levels[level][scale]
    (w1, b1), (w2, b2), (w3, b3) = sc.layers
    d = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    xyz_d, nx_d, bq_d = d(xyz), d(new_xyz), d(bq, torch.int32)
    f_d = None if feats is None else d(feats)
    cin = 0 if feats is None else feats.shape[2]
    st = stream_ptr()
    out_a = torch.zeros(B, npnt, cout_all, device="cuda")
    _lib.call("gp_sa_mlp_max", B, n, npnt, ns, cin, spec[0], spec[1], spec[2], ptr(xyz_d), ptr(f_d), ptr(nx_d), ptr(bq_d), ptr(w1), ptr(b1),
              ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(out_a), cout_all, off, st)
    out_b = torch.zeros(B, npnt, cout_all, device="cuda")
    z, zstride, zoff = None, 0, 0
    if cin:
        zstride = sum(s_.couts[0] for s_ in ew.levels[level])
        zoff = sum(s_.couts[0] for s_ in ew.levels[level][:scale])
        z = torch.empty(B, n, zstride, device="cuda")
        _lib.call("gp_point_linear", B * n, cin, zstride, ptr(f_d), ptr(ew.z_weights[level]), ptr(z), st)
    _lib.call("gp_sa_pre_mlp_max", B, n, npnt, ns, spec[0], spec[1], spec[2], ptr(xyz_d), ptr(nx_d), ptr(bq_d), ptr(z), zstride, zoff,
              ptr(sc.wxyz), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(out_b), cout_all, off, st)
    # ... and as the encoder calls it: with the hidden-layer layout the weights were packed for (level 2, 196 channels: the chain
    # kernels then skip the k-steps of the last 16-channel block that only hold padding)
    out_c = torch.zeros(B, npnt, cout_all, device="cuda")
    _lib.call("gp_sa_pre_mlp_max_layout", sc.hidden_layout, B, n, npnt, ns, spec[0], spec[1], spec[2], ptr(xyz_d), ptr(nx_d), ptr(bq_d), ptr(z),
              zstride, zoff, ptr(sc.wxyz), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(out_c), cout_all, off, st)
    assert sc.hidden_layout == (1 if spec[1] % 16 else 0)
    a = out_a[:, :, off:off + spec[-1]].cpu().numpy()
    b = out_b[:, :, off:off + spec[-1]].cpu().numpy()
    c = out_c[:, :, off:off + spec[-1]].cpu().numpy()
    np.testing.assert_allclose(a, ref, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(b, ref, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(c, ref, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(c, b, rtol=0, atol=2e-6 * np.abs(b).max())
    assert np.all(out_a[:, :, :off].cpu().numpy() == 0) and np.all(out_b[:, :, off + spec[-1]:].cpu().numpy() == 0)  # writes only its slice


def test_unsupported_shapes_fail_loudly():
    from genpose_amd import _lib
    from genpose_amd._lib import ptr, stream_ptr
    x = torch.zeros(1, 64, 3, device="cuda")
    idx = torch.zeros(1, 8, 24, dtype=torch.int32, device="cuda")
    w = torch.zeros(4096, device="cuda")
    out = torch.zeros(1, 8, 32, device="cuda")
    with pytest.raises(_lib.GenposeHipError):  # neighbourhoods are 8 samples or a multiple of 16 (every configuration of pointnet2.py:24-78)
        _lib.call("gp_sa_mlp_max", 1, 64, 8, 24, 0, 16, 16, 32, ptr(x), None, ptr(x), ptr(idx), ptr(w), ptr(w), ptr(w), ptr(w), ptr(w), ptr(w),
                  ptr(out), 32, 0, stream_ptr())


@pytest.mark.parametrize("ns", [8, 64])
def test_small_and_large_neighbourhoods_vs_oracle(ns):
    """nsample = 8 (two neighbourhoods per 16-row MFMA chunk, pooled over half rows) and 64, un-hoisted and hoisted tile kernels against
    the oracle's grouped shared MLP + max-pool, with input features."""
    from genpose_amd import _lib
    from genpose_amd._lib import ptr, stream_ptr
    from genpose_amd.weights import SAScale
    from oracle import pn2_oracle as ops
    B, n, npnt, cin, spec = 3, 200, 37, 32, [48, 80, 64]
    gen = torch.Generator().manual_seed(ns)
    xyz = torch.rand(B, n, 3, generator=gen) * 0.2
    feats = torch.randn(B, n, cin, generator=gen)
    new_xyz = xyz[:, :npnt].contiguous()
    bq = ops.ball_query(0.08, ns, xyz.numpy(), new_xyz.numpy())
    sd = {}
    chans = [cin + 3] + spec
    for l in range(3):
        p = f"m.layer{l}."
        sd[p + "conv.weight"] = torch.randn(chans[l + 1], chans[l], 1, 1, generator=gen) * (2.0 / chans[l]) ** 0.5
        sd[p + "bn.bn.weight"] = 1.0 + 0.1 * torch.randn(chans[l + 1], generator=gen)
        sd[p + "bn.bn.bias"] = 0.1 * torch.randn(chans[l + 1], generator=gen)
        sd[p + "bn.bn.running_mean"] = 0.1 * torch.randn(chans[l + 1], generator=gen)
        sd[p + "bn.bn.running_var"] = (1.0 + 0.1 * torch.randn(chans[l + 1], generator=gen)).abs() + 0.05
    g_xyz = torch.from_numpy(ops.group_points(np.ascontiguousarray(xyz.numpy().transpose(0, 2, 1)), bq)) - new_xyz.transpose(1, 2).unsqueeze(-1)
    g_f = torch.from_numpy(ops.group_points(np.ascontiguousarray(feats.numpy().transpose(0, 2, 1)), bq))
    ref = go._shared_mlp(sd, "m.", torch.cat([g_xyz, g_f], dim=1)).max(dim=3)[0].permute(0, 2, 1).numpy()  # [B, np, C]
    sc = SAScale(sd, "m.", cin, "cuda")
    (w1, b1), (w2, b2), (w3, b3) = sc.layers
    xd, fd, nd, idx = xyz.cuda(), feats.cuda(), new_xyz.cuda(), torch.from_numpy(bq).cuda()
    out_a = torch.zeros(B, npnt, spec[-1], device="cuda")
    _lib.call("gp_sa_mlp_max", B, n, npnt, ns, cin, *spec, ptr(xd), ptr(fd), ptr(nd), ptr(idx), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3),
              ptr(out_a), spec[-1], 0, stream_ptr())
    from genpose_amd.weights import pack_weight
    z = torch.empty(B, n, spec[0], device="cuda")
    _lib.call("gp_point_linear", B * n, cin, spec[0], ptr(fd), ptr(pack_weight(sc.w1_feat).cuda()), ptr(z), stream_ptr())
    out_b = torch.zeros(B, npnt, spec[-1], device="cuda")
    _lib.call("gp_sa_pre_mlp_max_layout", sc.hidden_layout, B, n, npnt, ns, *spec, ptr(xd), ptr(nd), ptr(idx), ptr(z), spec[0], 0, ptr(sc.wxyz), ptr(b1),
              ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(out_b), spec[-1], 0, stream_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out_a.cpu().numpy(), ref, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(out_b.cpu().numpy(), ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("k_in,n_out", [(96, 128), (256, 256), (512, 512), (64, 96)])
@pytest.mark.parametrize("rows", [1, 37, 4096 + 19])
def test_point_linear_vs_fp32_reference(k_in, n_out, rows):
    """gp_point_linear (hoisted feature half of a first layer: Z = X W^T) against the plain fp32 product, for the three shapes that
    run with the weights stationary in registers (ragged last tiles, fewer tiles than workgroups) and one that takes the generic
    kernel.  fp32 tolerance: 2e-6 of the largest output (k <= 512 products per output)."""
    from genpose_amd import _lib
    from genpose_amd._lib import ptr, stream_ptr
    from genpose_amd.weights import pack_weight
    gen = torch.Generator().manual_seed(rows + k_in)
    X = torch.randn(rows, k_in, generator=gen)
    W = torch.randn(n_out, k_in, generator=gen) / k_in ** 0.5
    wp = pack_weight(W).cuda()
    x_d = X.cuda()
    z = torch.full((rows + 1, n_out), 7.0, device="cuda")  # one guard row behind the output
    _lib.call("gp_point_linear", rows, k_in, n_out, ptr(x_d), ptr(wp), ptr(z), stream_ptr())
    torch.cuda.synchronize()
    ref = (X.double() @ W.double().T).float().numpy()
    got = z.cpu().numpy()
    assert (got[rows] == 7.0).all(), "wrote past the last row"
    np.testing.assert_allclose(got[:rows], ref, rtol=0, atol=2e-6 * np.abs(ref).max() * (k_in / 96) ** 0.5 + 1e-6)
