"""The RK45 driver's SHARED-CHUNK plan (GP_PLAN_SHARED, csrc/rk45.hip: rk45_attempt_shared_kernel): one workgroup per CU for the whole
attempt, the 16-row chunks that do not divide over the CUs handed from workgroup to workgroup across the six stages INSIDE the launch -
the plan scripts/eval_single.sh's batch shape takes (256 clouds x 50 candidates = 800 chunks on 256 CUs).

Checked: it is what the sampler picks at that shape (and only where it applies); against the whole-tile plans on the same inputs - same
accept / reject sequence, same evaluation count, poses to 1e-5 (observed 1e-13: every row's arithmetic is the whole-tile plans' bit for bit,
only the error norm's partial sums add in another order); repeatable bit for bit (the hand-over is ordered, not racy); the trajectory record;
the 16-row-own-tile variant (4 097 - 4 768 rows); the sizing query and the plan-0 contract of the C boundary.  Against the CPU oracle the
shape runs in tests/test_gpu_fullsize.py::test_drop_in_eval_single_as_timed and tests/test_gpu_trained_regime.py."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import genpose_oracle as go

SHARED, HEADSPLIT = 0x200, 0x100


@pytest.fixture(scope="module")
def snet():
    from genpose_amd.scorenet import ScoreNetHIP
    return ScoreNetHIP(go.make_state_dict(0, "score"), "cuda")


def _inputs(B, K, seed):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 768, generator=gen).cuda(), torch.randn(B, 3, generator=gen).cuda(), torch.randn(B * K, 9, generator=gen).cuda())


def _solve(smp, cvec, centre, x0, T0=0.55, **kw):
    out = smp.run(cvec, centre, x0, T0=T0, **kw)
    st = smp.last_stats
    return out, (int(st["nfev"]), int(st["n_attempts"]), [bool(a) for a in st["log_acc"]], int(st["status"]))


@pytest.mark.parametrize("B,K,own,other", [(256, 50, 48, 64), (90, 50, 16, 32), (250, 50, 48, 32)])
def test_shared_chunk_plan_against_whole_tiles(snet, B, K, own, other):
    from genpose_amd import _lib
    from genpose_amd.samplers import ODESampler
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    if ncu != 256:
        pytest.skip(f"shape table written for 256 CUs, device has {ncu}")
    cvec, centre, x0 = _inputs(B, K, B)
    a = ODESampler(snet, B, K, "cuda")
    assert a.plan == (own | SHARED) and a.shared and a.tile == own
    b = ODESampler(snet, B, K, "cuda", tile=other)
    assert b.plan == other and not b.shared
    (_, xa), sa = _solve(a, cvec, centre, x0)
    (_, xb), sb = _solve(b, cvec, centre, x0)
    assert sa[3] == 1 and sb[3] == 1
    # the same controller decisions (the error norm's partial sums add up in another order: ulp-level differences in the step sizes)
    assert abs(sa[0] - sb[0]) <= 6 and abs(sa[1] - sb[1]) <= 1, (sa[:2], sb[:2])
    common = min(len(sa[2]), len(sb[2]))
    assert sa[2][:common] == sb[2][:common]
    xa, xb = xa.cpu().numpy(), xb.cpu().numpy()
    assert np.isfinite(xa).all()
    d_rot, d_tr = np.abs(xa[:, :6] - xb[:, :6]).max(), np.abs(xa[:, 6:] - xb[:, 6:]).max() / np.abs(xb[:, 6:]).max()
    print(f"{B} x {K}: shared-chunk plan {own} vs whole tiles {other}: nfev {sa[0]} / {sb[0]}, max rotation diff {d_rot:.2e}, translation (rel) {d_tr:.2e}")
    # The four-wave tiles of this plan form the 256 -> 3 output sums in the ORDER of the eight-wave 32- / 64-row tiles (score_trunk.h: ORDER8), so
    # every row's right-hand side is the whole-tile plans' bit for bit; what remains is the order in which the error norm's partial sums are
    # added (a step size that differs in its last bit): two WHOLE-tile plans of one problem differ by the same 1e-15 .. 1e-6
    # (scratch/shared_plan_control.py: 64- vs 32-row tiles median 0, max 3.7e-6 over 100 inputs; this plan vs 32-row tiles max 8.5e-14)
    assert d_rot < 1e-5 and d_tr < 1e-7, (d_rot, d_tr)
    # the rows of the SHARED chunks in particular (they are the ones that cross workgroups): same bound, looked at on their own
    first_shared = ncu * own
    assert np.abs(xa[first_shared:, :6] - xb[first_shared:, :6]).max() < 1e-5
    # repeatable bit for bit - also through the graph replays of the steady state
    for _ in range(3):
        (_, xr), sr = _solve(a, cvec, centre, x0)
        assert sr == sa and np.array_equal(xr.cpu().numpy(), xa)


def test_shared_chunk_plan_records_the_trajectory(snet):
    """return_process (accepted states) and sampling_steps (dense output at t_eval) read y / y_new / K of the shared chunks in LATER launches."""
    from genpose_amd.samplers import ODESampler
    B, K = 256, 50
    cvec, centre, x0 = _inputs(B, K, 77)
    a, b = ODESampler(snet, B, K, "cuda"), ODESampler(snet, B, K, "cuda", tile=64)
    if not a.shared:
        pytest.skip("shared-chunk plan does not apply on this device")
    for kw in (dict(return_process=True), dict(return_process=True, num_steps=20)):
        (pa, xa), sa = _solve(a, cvec, centre, x0, **kw)
        (pb, xb), sb = _solve(b, cvec, centre, x0, **kw)
        assert sa[0] == sb[0] and pa.shape == pb.shape
        np.testing.assert_allclose(pa.cpu().numpy()[..., :6], pb.cpu().numpy()[..., :6], rtol=0, atol=1e-5)
        scale = float(pb[..., 6:].abs().max())
        np.testing.assert_allclose(pa.cpu().numpy()[..., 6:], pb.cpu().numpy()[..., 6:], rtol=0, atol=1e-7 * scale)


def test_plan_queries_of_the_c_boundary():
    from genpose_amd import _lib
    L = _lib.lib()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    # latency regime: the recommendation is head-split, plan 0 resolves to whole tiles (a caller's 3 x tiles buffer stays sufficient)
    assert L.gp_rk45_plan_rows(0, 1, 1, 10) == (16 | HEADSPLIT)
    assert L.gp_rk45_partials_count(0, 0, 1, 1, 10) == 3 and L.gp_rk45_partials_count(0, 16 | HEADSPLIT, 1, 1, 10) == 9
    assert L.gp_rk45_partials_count(0, 16, 1, 5, 50) == 3 * 16 and L.gp_rk45_partials_count(0, 0, 10, 64, 50) > 0
    if ncu == 256:
        assert L.gp_rk45_plan_rows(0, 1, 256, 50) == (48 | SHARED) and L.gp_rk45_plan_rows_unshared(0, 1, 256, 50) == 64
        assert L.gp_rk45_partials_count(0, 48 | SHARED, 1, 256, 50) == 3 * max(200, 256 + 32)
        assert L.gp_rk45_plan_rows(0, 1, 64, 50) in (16, 32) and L.gp_rk45_plan_rows(0, 2, 256, 50) != (48 | SHARED)  # 200 chunks; two groups
        assert L.gp_rk45_partials_count(0, 48 | SHARED, 1, 64, 50) == -1 and L.gp_rk45_partials_count(1, 48 | SHARED, 1, 256, 50) == -1
        # the energy model's score and the likelihood ODE never take it
        assert L.gp_rk45_plan_rows(1, 1, 256, 50) in (16, 128) and L.gp_rk45_plan_rows(2, 1, 256, 50) in (16, 128)


def test_shared_chunk_plan_beside_a_busy_stream(snet):
    """The hand-over inside the launch must not depend on all 256 workgroups being resident at once: with another stream's kernels owning CUs
    (an encoder pass of 256 clouds: persistent workgroups on every CU) the attempt's workgroups start in waves, a consumer of a shared chunk
    waits (bounded) for a producer that has a LOWER workgroup index - dispatched before it - and the result must be the undisturbed run's,
    bit for bit, with no solve failing on the bounded wait."""
    from genpose_amd import synth
    from genpose_amd.encoder import Pointnet2EncoderHIP
    from genpose_amd.samplers import ODESampler
    B, K = 256, 50
    cvec, centre, x0 = _inputs(B, K, 5)
    smp = ODESampler(snet, B, K, "cuda")
    if not smp.shared:
        pytest.skip("shared-chunk plan does not apply on this device")
    (_, quiet), sq = _solve(smp, cvec, centre, x0)
    quiet = quiet.clone()
    enc = Pointnet2EncoderHIP(go.make_state_dict(0, "score"), "cuda")
    big = torch.from_numpy(synth.make_batch(256, start=5000)).cuda()
    side = torch.cuda.Stream()
    enc.forward(big)
    torch.cuda.synchronize()
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(4):
                enc.forward(big)  # ~15 ms of kernels that fill every CU, underneath the 14 ms solve
        (_, got), sg = _solve(smp, cvec, centre, x0)
        assert sg == sq and torch.equal(got, quiet), f"solve {rep} beside a busy stream"
    torch.cuda.synchronize()
