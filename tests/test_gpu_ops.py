"""GPU parity: section A of the C ABI (pointnet2_cuda operator surface) against the C oracle and the golden vectors.
Integer/index outputs must be BIT-EXACT."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pn2_oracle as ops


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def pn2():
    import genpose_amd.pointnet2_cuda as m
    return m


@pytest.fixture(params=["A", "B", "C"])
def arith(request, pn2):
    """Every contraction convention of the three-product sums (include/genpose_hip.h GP_ARITH_*): the operator module and the oracle
    are switched together, so each test below holds the HIP kernels to the oracle bit for bit under all three."""
    from genpose_amd.config import DEFAULT_DIST_ARITH
    pn2.set_arith(request.param)
    try:
        with ops.use_arith(request.param):
            yield request.param
    finally:
        pn2.set_arith(DEFAULT_DIST_ARITH)


def golden_name(arith):
    return "g1_g2_ops.npz" if arith == ops.DEFAULT_ARITH else f"g1_g2_ops_arith{arith}.npz"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def hip_fps(pn2, xyz, m, temp=None):
    B, N, _ = xyz.shape
    x = dev(xyz)
    t = torch.full((B, N), 1e10, device="cuda") if temp is None else dev(temp)
    idx = torch.empty(B, m, dtype=torch.int32, device="cuda")
    assert pn2.furthest_point_sampling_wrapper(B, N, m, x, t, idx) == 1
    return idx.cpu().numpy(), t.cpu().numpy()


def hip_bq(pn2, r, ns, xyz, new_xyz):
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.zeros(B, M, ns, dtype=torch.int32, device="cuda")
    pn2.ball_query_wrapper(B, N, M, r, ns, dev(new_xyz), dev(xyz), idx)
    return idx.cpu().numpy()


def test_fps_bq_golden(pn2, golden, arith):
    g = golden(golden_name(arith))
    assert str(g["arith"]) == arith
    cur = g["clouds"]
    radii = [[0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]
    for lvl, npnt in enumerate([512, 256, 128]):
        idx, _ = hip_fps(pn2, cur, npnt)
        assert np.array_equal(idx, g[f"fps_idx{lvl}"].astype(np.int32)), f"FPS level {lvl}"
        new = np.take_along_axis(cur, idx[:, :, None].astype(np.int64), axis=1)
        for s, ns in enumerate([16, 32]):
            bq = hip_bq(pn2, radii[lvl][s], ns, cur, new)
            assert np.array_equal(bq[0], g[f"bq{lvl}_{s}_cloud0"].astype(np.int32))
            assert [sha(bq[b]) for b in range(bq.shape[0])] == list(g[f"bq{lvl}_{s}_sha"])
        cur = np.ascontiguousarray(new)
    odd = g["odd_clouds"]
    assert np.array_equal(hip_fps(pn2, odd, 100)[0], g["odd_fps"])
    assert np.array_equal(hip_bq(pn2, 0.05, 5, odd, np.ascontiguousarray(odd[:, :50])), g["odd_bq"])
    # the other two sites of the contraction (interpolate_gpu.cu:36,95), captured through the reference's three_nn / three_interpolate
    unk, kn = np.ascontiguousarray(g["clouds"][:2, :300]), np.ascontiguousarray(g["clouds"][:2, 300:364])
    d2 = torch.empty(2, 300, 3, device="cuda")
    i3 = torch.empty(2, 300, 3, dtype=torch.int32, device="cuda")
    pn2.three_nn_wrapper(2, 300, 64, dev(unk), dev(kn), d2, i3)
    assert np.array_equal(i3.cpu().numpy(), g["nn_idx"].astype(np.int32))
    assert np.array_equal(d2.cpu().numpy(), g["nn_dist2"])
    out = torch.empty(2, 7, 300, device="cuda")
    pn2.three_interpolate_wrapper(2, 7, 64, 300, dev(g["interp_feats"]), i3, dev(g["interp_w"]), out)
    assert np.array_equal(out.cpu().numpy(), g["interp_out"])


def test_conventions_pick_different_centres(pn2):
    """The switch is not a no-op: on the golden clouds (exact grid ties, tiled duplicates) the three conventions select different points,
    and the HIP kernels follow the oracle into each of them (test_fps_bq_golden); here: the default is B, and A / C differ from it."""
    from genpose_amd.config import DEFAULT_DIST_ARITH
    assert pn2.ARITH == DEFAULT_DIST_ARITH == ops.DEFAULT_ARITH == "B"
    from tests.conftest import load_golden
    clouds = load_golden("g1_g2_ops.npz")["clouds"]
    got = {}
    for a in "ABC":
        pn2.set_arith(a)
        try:
            got[a] = hip_fps(pn2, clouds, 512)[0]
        finally:
            pn2.set_arith(DEFAULT_DIST_ARITH)
    assert not np.array_equal(got["A"], got["B"]) and not np.array_equal(got["B"], got["C"]) and not np.array_equal(got["A"], got["C"])


@pytest.mark.parametrize("n,m", [(1024, 512), (512, 256), (256, 128), (700, 100), (64, 64), (65, 7), (3, 2), (1, 1), (1500, 300),
                                 (4096, 64), (5000, 40)])
def test_fps_vs_oracle(pn2, arith, n, m):
    rng = np.random.default_rng(n * 7 + m)
    xyz = (rng.normal(size=(3, n, 3)) * 0.1).astype(np.float32)
    xyz[1] = np.round(xyz[1] * 50) / 50  # many exact ties
    if n > 8:
        xyz[2, n // 2:] = xyz[2, : n - n // 2]  # exact duplicates (tiled clouds)
    ref, ref_t = ops.furthest_point_sampling(xyz, m)
    got, got_t = hip_fps(pn2, xyz, m)
    assert np.array_equal(got, ref)
    assert np.array_equal(got_t, ref_t)  # running min distances persist bit-exactly


def test_fps_respects_temp_and_ties(pn2, arith):
    p = np.array([[[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0.5, 0, 0]]], dtype=np.float32)
    assert hip_fps(pn2, p, 3)[0].tolist() == [[0, 2, 1]]
    q = np.zeros((1, 8, 3), dtype=np.float32)
    q[0, 1:] = [3, 0, 0]
    assert hip_fps(pn2, q, 2)[0].tolist() == [[0, 4]]
    rng = np.random.default_rng(3)
    xyz = rng.normal(size=(2, 300, 3)).astype(np.float32)
    temp = rng.uniform(0, 0.5, size=(2, 300)).astype(np.float32)
    ref, ref_t = ops.furthest_point_sampling(xyz, 50, temp.copy())
    got, got_t = hip_fps(pn2, xyz, 50, temp.copy())
    assert np.array_equal(got, ref) and np.array_equal(got_t, ref_t)


@pytest.mark.parametrize("n,m,r,ns", [(1024, 512, 0.02, 16), (1024, 512, 0.04, 32), (512, 256, 0.08, 32), (256, 128, 0.16, 32),
                                      (700, 50, 0.05, 5), (100, 100, 10.0, 64), (70, 3, 1e-6, 8), (6000, 33, 0.1, 20)])
def test_ball_query_vs_oracle(pn2, arith, n, m, r, ns):
    rng = np.random.default_rng(n + m)
    xyz = (rng.normal(size=(2, n, 3)) * 0.08).astype(np.float32)
    new = np.ascontiguousarray(xyz[:, rng.permutation(n)[:m]])
    new[:, 0] += 100.0  # a centre with no neighbour at all: row must stay zero
    ref = ops.ball_query(r, ns, xyz, new)
    got = hip_bq(pn2, r, ns, xyz, new)
    assert np.array_equal(got, ref)
    assert np.all(got[:, 0] == 0)


def test_ball_query_radius_tie_is_strict(pn2):
    # points exactly at distance r: d2 < r^2 is strict (ball_query_gpu.cu:34)
    xyz = np.array([[[0, 0, 0], [0.5, 0, 0], [0.25, 0, 0], [0, 0.5, 0]]], dtype=np.float32)
    new = np.array([[[0, 0, 0]]], dtype=np.float32)
    assert hip_bq(pn2, 0.5, 4, xyz, new).tolist() == ops.ball_query(0.5, 4, xyz, new).tolist() == [[[0, 2, 0, 0]]]


def test_msg_ball_query_and_chain(golden, arith):
    """Fused entry points used by the encoder: gp_fps_chain / gp_ball_query_msg (+ their `_arith` twins: every convention)."""
    import ctypes
    from genpose_amd import _lib
    from genpose_amd._lib import ptr, stream_ptr
    from genpose_amd.config import dist_arith_code
    g = golden(golden_name(arith))
    ac = dist_arith_code(arith)
    xyz = dev(g["clouds"])
    B = xyz.shape[0]
    ms = [512, 256, 128]
    idx = [torch.empty(B, m, dtype=torch.int32, device="cuda") for m in ms]
    nx = [torch.empty(B, m, 3, device="cuda") for m in ms]
    _lib.call("gp_fps_chain_arith", ac, B, 1024, 3, (ctypes.c_int * 3)(*ms), ptr(xyz), ptr(idx[0]), ptr(nx[0]), ptr(idx[1]), ptr(nx[1]),
              ptr(idx[2]), ptr(nx[2]), stream_ptr())
    if arith == ops.DEFAULT_ARITH:  # the un-suffixed entry points ARE the default convention
        assert _lib.lib().gp_arith_default() == ac
        idx_d = [torch.empty_like(t) for t in idx]
        nx_d = [torch.empty_like(t) for t in nx]
        _lib.call("gp_fps_chain", B, 1024, 3, (ctypes.c_int * 3)(*ms), ptr(xyz), ptr(idx_d[0]), ptr(nx_d[0]), ptr(idx_d[1]), ptr(nx_d[1]),
                  ptr(idx_d[2]), ptr(nx_d[2]), stream_ptr())
        assert all(torch.equal(a, b) for a, b in zip(idx + nx, idx_d + nx_d))
    cur = g["clouds"]
    for l in range(3):
        assert np.array_equal(idx[l].cpu().numpy(), g[f"fps_idx{l}"].astype(np.int32))
        new = np.take_along_axis(cur, g[f"fps_idx{l}"].astype(np.int64)[:, :, None], axis=1)
        assert np.array_equal(nx[l].cpu().numpy(), new)
        i0 = torch.full((B, ms[l], 16), -7, dtype=torch.int32, device="cuda")
        i1 = torch.full((B, ms[l], 32), -7, dtype=torch.int32, device="cuda")
        r = [[0.02, 0.04], [0.04, 0.08], [0.08, 0.16]][l]
        _lib.call("gp_ball_query_msg_arith", ac, B, cur.shape[1], ms[l], r[0], 16, r[1], 32, ptr(nx[l]), ptr(dev(cur)), ptr(i0), ptr(i1), stream_ptr())
        assert np.array_equal(i0.cpu().numpy(), ops.ball_query(r[0], 16, cur, new))
        assert np.array_equal(i1.cpu().numpy(), ops.ball_query(r[1], 32, cur, new))
        cur = np.ascontiguousarray(new)


def test_gather_group(pn2):
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(3, 7, 200)).astype(np.float32)
    idx = rng.integers(0, 200, size=(3, 40)).astype(np.int32)
    out = torch.empty(3, 7, 40, device="cuda")
    pn2.gather_points_wrapper(3, 7, 200, 40, dev(pts), dev(idx), out)
    assert np.array_equal(out.cpu().numpy(), ops.gather_points(pts, idx))
    gidx = rng.integers(0, 200, size=(3, 40, 9)).astype(np.int32)
    out = torch.empty(3, 7, 40, 9, device="cuda")
    pn2.group_points_wrapper(3, 7, 200, 40, 9, dev(pts), dev(gidx), out)
    assert np.array_equal(out.cpu().numpy(), ops.group_points(pts, gidx))
    # backward scatters (atomic accumulation order differs -> tolerance)
    go_ = rng.normal(size=(3, 7, 40, 9)).astype(np.float32)
    gp = torch.zeros(3, 7, 200, device="cuda")
    pn2.group_points_grad_wrapper(3, 7, 200, 40, 9, dev(go_), dev(gidx), gp)
    np.testing.assert_allclose(gp.cpu().numpy(), ops.group_points_grad(go_, gidx, 200), rtol=1e-5, atol=1e-5)
    go2 = rng.normal(size=(3, 7, 40)).astype(np.float32)
    gp = torch.zeros(3, 7, 200, device="cuda")
    pn2.gather_points_grad_wrapper(3, 7, 200, 40, dev(go2), dev(idx), gp)
    np.testing.assert_allclose(gp.cpu().numpy(), ops.gather_points_grad(go2, idx, 200), rtol=1e-5, atol=1e-5)


def test_three_nn_interpolate(pn2, arith):
    rng = np.random.default_rng(9)
    for (n, m) in [(300, 64), (50, 3), (1000, 3000)]:
        unk = rng.normal(size=(2, n, 3)).astype(np.float32)
        kn = rng.normal(size=(2, m, 3)).astype(np.float32)
        kn[0, 1] = kn[0, 0]  # exact tie: strict '<' keeps the earlier index first
        d = torch.empty(2, n, 3, device="cuda")
        i = torch.empty(2, n, 3, dtype=torch.int32, device="cuda")
        pn2.three_nn_wrapper(2, n, m, dev(unk), dev(kn), d, i)
        rd, ri = ops.three_nn(unk, kn)
        assert np.array_equal(i.cpu().numpy(), ri)
        assert np.array_equal(d.cpu().numpy(), rd)
        feats = rng.normal(size=(2, 5, m)).astype(np.float32)
        w = rng.uniform(size=(2, n, 3)).astype(np.float32)
        out = torch.empty(2, 5, n, device="cuda")
        pn2.three_interpolate_wrapper(2, 5, m, n, dev(feats), dev(ri), dev(w), out)
        assert np.array_equal(out.cpu().numpy(), ops.three_interpolate(feats, ri, w))
        g = rng.normal(size=(2, 5, n)).astype(np.float32)
        gp = torch.zeros(2, 5, m, device="cuda")
        pn2.three_interpolate_grad_wrapper(2, 5, n, m, dev(g), dev(ri), dev(w), gp)
        np.testing.assert_allclose(gp.cpu().numpy(), ops.three_interpolate_grad(g, ri, w, m), rtol=1e-4, atol=1e-4)


def test_reference_error_behaviour(pn2):
    x = torch.zeros(1, 8, 3, device="cuda")
    with pytest.raises(RuntimeError):  # ball_query.cpp:12-21 CHECK_INPUT
        pn2.ball_query_wrapper(1, 8, 8, 0.1, 4, x.cpu(), x, torch.zeros(1, 8, 4, dtype=torch.int32, device="cuda"))
    with pytest.raises(RuntimeError):
        pn2.ball_query_wrapper(1, 8, 8, 0.1, 4, x.transpose(1, 2), x, torch.zeros(1, 8, 4, dtype=torch.int32, device="cuda"))


def test_pose_to_RT_kernels_match_the_tensor_formulas():
    """gp_pose9_to_rt / gp_quat_trans_to_rt (one launch each; what the runners hand on as 4x4 matrices) against the same helpers on host
    tensors (get_rot_matrix in float64, pytorch3d's quaternion_to_matrix formula)."""
    from genpose_amd import rotation
    gen = torch.Generator().manual_seed(3)
    pose = torch.randn(7, 13, 9, generator=gen, dtype=torch.float64)
    pose[0, 0, :3] = 0.0  # a degenerate first column: F.normalize's eps path
    for p in (pose, pose.float()):
        got = rotation.pose9_to_RT(p.cuda())
        ref = rotation.pose9_to_RT(p)
        assert got.dtype == torch.float64 and got.shape == (7, 13, 4, 4)
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-12 if p.dtype == torch.float64 else 1e-12)
    qt = torch.randn(33, 7, generator=gen)
    got = rotation.quat_trans_to_RT(qt.cuda())
    ref = rotation.quat_trans_to_RT(qt)
    assert got.dtype == torch.float32 and got.shape == (33, 4, 4)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6 * float(ref.abs().max()))
    assert torch.equal(got[:, 3].cpu(), torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(33, 4))
