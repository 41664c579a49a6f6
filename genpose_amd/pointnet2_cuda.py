"""Drop-in for the reference's CUDA extension module `pointnet2_cuda`
(networks/pts_encoder/pointnet2_utils/pointnet2/src/pointnet2_api.cpp:10-24): the same nine function names and
positional signatures, operating in place on caller-allocated, contiguous device tensors and launching on
torch's current stream - backed by libgenpose_hip.so (hand-written gfx950 kernels).

Usage inside the reference tree:   import genpose_amd.pointnet2_cuda as pointnet2_cuda
(or `sys.modules["pointnet2_cuda"] = genpose_amd.pointnet2_cuda` before importing pointnet2_utils.py).

Error behaviour: like ball_query.cpp:12-21 (TORCH_CHECK -> RuntimeError) every tensor must be a contiguous
device tensor; unlike the reference launchers (fprintf + exit(-1)) a failed launch raises GenposeHipError.
"""
import torch

from . import _lib
from ._lib import ptr, stream_ptr
from .config import DEFAULT_DIST_ARITH, dist_arith_code

# Contraction convention of the three-product sums in furthest point sampling, ball query, three-NN and three-interpolate
# ('A' | 'B' | 'C', include/genpose_hip.h GP_ARITH_*).  The reference's nine signatures have no room for it: it is module state.
ARITH = DEFAULT_DIST_ARITH


def set_arith(arith):
    global ARITH
    dist_arith_code(arith)
    ARITH = arith


def _chk(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("tensor must be a CUDAtensor ")
        if not t.is_contiguous():
            raise RuntimeError("tensor must be contiguous ")


def _f32(*ts):
    for t in ts:
        if t.dtype != torch.float32:
            raise RuntimeError("expected a float32 tensor")


def _i32(*ts):
    for t in ts:
        if t.dtype != torch.int32:
            raise RuntimeError("expected an int32 tensor")


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    _chk(new_xyz, xyz, idx); _f32(new_xyz, xyz); _i32(idx)
    _lib.check_device()
    _lib.call("gp_ball_query_arith", dist_arith_code(ARITH), b, n, m, float(radius), nsample, ptr(new_xyz), ptr(xyz), ptr(idx), stream_ptr())
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _chk(points, idx, out); _f32(points, out); _i32(idx)
    _lib.check_device()
    _lib.call("gp_group_points", b, c, n, npoints, nsample, ptr(points), ptr(idx), ptr(out), stream_ptr())
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _chk(grad_out, idx, grad_points); _f32(grad_out, grad_points); _i32(idx)
    _lib.check_device()
    _lib.call("gp_group_points_grad", b, c, n, npoints, nsample, ptr(grad_out), ptr(idx), ptr(grad_points), stream_ptr())
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _chk(points, idx, out); _f32(points, out); _i32(idx)
    _lib.check_device()
    _lib.call("gp_gather_points", b, c, n, npoints, ptr(points), ptr(idx), ptr(out), stream_ptr())
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _chk(grad_out, idx, grad_points); _f32(grad_out, grad_points); _i32(idx)
    _lib.check_device()
    _lib.call("gp_gather_points_grad", b, c, n, npoints, ptr(grad_out), ptr(idx), ptr(grad_points), stream_ptr())
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    _chk(points, temp, idx); _f32(points, temp); _i32(idx)
    _lib.check_device()
    _lib.call("gp_furthest_point_sampling_arith", dist_arith_code(ARITH), b, n, m, ptr(points), ptr(temp), ptr(idx), stream_ptr())
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _chk(unknown, known, dist2, idx); _f32(unknown, known, dist2); _i32(idx)
    _lib.check_device()
    _lib.call("gp_three_nn_arith", dist_arith_code(ARITH), b, n, m, ptr(unknown), ptr(known), ptr(dist2), ptr(idx), stream_ptr())
    return 1


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _chk(points, idx, weight, out); _f32(points, weight, out); _i32(idx)
    _lib.check_device()
    _lib.call("gp_three_interpolate_arith", dist_arith_code(ARITH), b, c, m, n, ptr(points), ptr(idx), ptr(weight), ptr(out), stream_ptr())


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _chk(grad_out, idx, weight, grad_points); _f32(grad_out, weight, grad_points); _i32(idx)
    _lib.check_device()
    _lib.call("gp_three_interpolate_grad", b, c, n, m, ptr(grad_out), ptr(idx), ptr(weight), ptr(grad_points), stream_ptr())
