"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/pn2_ops.c header).

ctypes binding of oracle/build/libpn2_oracle.so: numpy-in / numpy-out CPU versions of the
reference's `pointnet2_cuda` kernels.  Also exposes `as_pointnet2_cuda_module()`, a CPU object
with the 9 pybind names of pointnet2_api.cpp:10-24 operating in place on CPU torch tensors, which
the import shim (oracle/ref_import.py) installs as `pointnet2_cuda` so that the reference's own
Python can run in the build container.
"""
import contextlib
import ctypes
import os
import subprocess
import numpy as np

# How `a*a + b*b + c*c` (sampling_gpu.cu:133, ball_query_gpu.cu:33, interpolate_gpu.cu:36,95) is contracted - see oracle/pn2_ops.c.
#   "A" fma(c,c, fma(b,b, a*a))   "B" fma(c,c, fma(a,a, b*b)) (what LLVM and GCC emit for the text; default)   "C" no contraction
ARITH_CODES = {"A": 0, "B": 1, "C": 2}
DEFAULT_ARITH = "B"   # must equal genpose_amd.config.DEFAULT_DIST_ARITH (tests/test_abi_and_host.py asserts it)
_current = [DEFAULT_ARITH]


def current_arith():
    return _current[-1]


@contextlib.contextmanager
def use_arith(arith):
    """Every oracle call inside the block (also through genpose_oracle / the imported reference's stand-in module) uses `arith`
    unless it names its own."""
    code(arith)
    _current.append(arith)
    try:
        yield
    finally:
        _current.pop()


def code(arith=None):
    a = _current[-1] if arith is None else arith
    if a not in ARITH_CODES:
        raise ValueError(f"arith must be one of {sorted(ARITH_CODES)}, got {a!r}")
    return ARITH_CODES[a]

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libpn2_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "pn2_ops.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def opt_n_threads(n):
    return lib().gpo_opt_n_threads(int(n))


def sqdist(a, b, arith=None):
    """One squared distance |a - b|^2 of two float32 3-vectors under `arith`."""
    a, pa = _f(a)
    b, pb = _f(b)
    fn = lib().gpo_sqdist
    fn.restype = ctypes.c_float
    return np.float32(fn(code(arith), pa, pb))


def furthest_point_sampling(xyz, m, temp=None, arith=None):
    """xyz (B,N,3) f32 -> idx (B,m) i32; temp (B,N) starts at 1e10 (pointnet2_utils.py:27)."""
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    if temp is None:
        temp = np.full((B, N), 1e10, dtype=np.float32)
    temp, pt = _f(temp)
    idx = np.zeros((B, m), dtype=np.int32)
    lib().gpo_furthest_point_sampling(code(arith), B, N, int(m), px, pt, idx.ctypes.data_as(ctypes.c_void_p))
    return idx, temp


def gather_points(points, idx):
    """points (B,C,N), idx (B,M) -> (B,C,M)"""
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    M = idx.shape[1]
    out = np.empty((B, C, M), dtype=np.float32)
    lib().gpo_gather_points(B, C, N, M, pp, pi, out.ctypes.data_as(ctypes.c_void_p))
    return out


def ball_query(radius, nsample, xyz, new_xyz, arith=None):
    """xyz (B,N,3), new_xyz (B,M,3) -> idx (B,M,nsample) i32 (pre-zeroed as pointnet2_utils.py:219)."""
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), dtype=np.int32)
    lib().gpo_ball_query(code(arith), B, N, M, ctypes.c_float(radius), int(nsample), pn, px, idx.ctypes.data_as(ctypes.c_void_p))
    return idx


def group_points(points, idx):
    """points (B,C,N), idx (B,np,ns) -> (B,C,np,ns)"""
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    _, npnt, ns = idx.shape
    out = np.empty((B, C, npnt, ns), dtype=np.float32)
    lib().gpo_group_points(B, C, N, npnt, ns, pp, pi, out.ctypes.data_as(ctypes.c_void_p))
    return out


def three_nn(unknown, known, arith=None):
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, N, _ = unknown.shape
    M = known.shape[1]
    d = np.empty((B, N, 3), dtype=np.float32)
    i = np.empty((B, N, 3), dtype=np.int32)
    lib().gpo_three_nn(code(arith), B, N, M, pu, pk, d.ctypes.data_as(ctypes.c_void_p), i.ctypes.data_as(ctypes.c_void_p))
    return d, i


def three_interpolate(points, idx, weight, arith=None):
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, M = points.shape
    N = idx.shape[1]
    out = np.empty((B, C, N), dtype=np.float32)
    lib().gpo_three_interpolate(code(arith), B, C, M, N, pp, pi, pw, out.ctypes.data_as(ctypes.c_void_p))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, npnt, ns = grad_out.shape
    out = np.zeros((B, C, n), dtype=np.float32)
    lib().gpo_group_points_grad(B, C, n, npnt, ns, pg, pi, out.ctypes.data_as(ctypes.c_void_p))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M = grad_out.shape
    out = np.zeros((B, C, n), dtype=np.float32)
    lib().gpo_gather_points_grad(B, C, n, M, pg, pi, out.ctypes.data_as(ctypes.c_void_p))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, N = grad_out.shape
    out = np.zeros((B, C, m), dtype=np.float32)
    lib().gpo_three_interpolate_grad(B, C, N, m, pg, pi, pw, out.ctypes.data_as(ctypes.c_void_p))
    return out


class _Pointnet2CudaCPU:
    """CPU object with the pybind surface of pointnet2_api.cpp:10-24 (in-place on CPU torch tensors).  The arithmetic convention
    is the oracle's current one (use_arith) at the time of each call."""

    @staticmethod
    def _p(t):
        assert t.is_contiguous()
        return ctypes.c_void_p(t.data_ptr())

    def furthest_point_sampling_wrapper(self, b, n, m, points, temp, idx):
        lib().gpo_furthest_point_sampling(code(), b, n, m, self._p(points), self._p(temp), self._p(idx))
        return 1

    def gather_points_wrapper(self, b, c, n, npoints, points, idx, out):
        lib().gpo_gather_points(b, c, n, npoints, self._p(points), self._p(idx), self._p(out))
        return 1

    def ball_query_wrapper(self, b, n, m, radius, nsample, new_xyz, xyz, idx):
        lib().gpo_ball_query(code(), b, n, m, ctypes.c_float(radius), nsample, self._p(new_xyz), self._p(xyz), self._p(idx))
        return 1

    def group_points_wrapper(self, b, c, n, npoints, nsample, points, idx, out):
        lib().gpo_group_points(b, c, n, npoints, nsample, self._p(points), self._p(idx), self._p(out))
        return 1

    def three_nn_wrapper(self, b, n, m, unknown, known, dist2, idx):
        lib().gpo_three_nn(code(), b, n, m, self._p(unknown), self._p(known), self._p(dist2), self._p(idx))
        return 1

    def three_interpolate_wrapper(self, b, c, m, n, points, idx, weight, out):
        lib().gpo_three_interpolate(code(), b, c, m, n, self._p(points), self._p(idx), self._p(weight), self._p(out))

    def group_points_grad_wrapper(self, b, c, n, npoints, nsample, grad_out, idx, grad_points):
        lib().gpo_group_points_grad(b, c, n, npoints, nsample, self._p(grad_out), self._p(idx), self._p(grad_points))
        return 1

    def gather_points_grad_wrapper(self, b, c, n, npoints, grad_out, idx, grad_points):
        lib().gpo_gather_points_grad(b, c, n, npoints, self._p(grad_out), self._p(idx), self._p(grad_points))
        return 1

    def three_interpolate_grad_wrapper(self, b, c, n, m, grad_out, idx, weight, grad_points):
        lib().gpo_three_interpolate_grad(b, c, n, m, self._p(grad_out), self._p(idx), self._p(weight), self._p(grad_points))


def as_pointnet2_cuda_module():
    return _Pointnet2CudaCPU()
