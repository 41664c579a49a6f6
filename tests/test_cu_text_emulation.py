"""Build container only (needs /root/reference, g++ and an FMA-capable host): a DIAGNOSTIC, not a parity pin.

The TEXT of the reference's CUDA kernels - sampling_gpu.cu (furthest point sampling incl. its shared-memory tree), ball_query_gpu.cu,
interpolate_gpu.cu (three_nn, three_interpolate) - is extracted at test time into a temporary directory, compiled by the HOST compilers
against oracle/cu_emul/cuda_model.h (every CUDA thread a cooperative fiber, __syncthreads() a round-robin barrier) and run.  Nothing of
the reference is committed or shipped; the model stands in for the CUDA headers, so this is not an oracle/_ref build.

What it establishes:
  1. oracle/pn2_ops.c follows the kernel text (tie rule of the tree, multi-point threads, first-hit prefill, double best-distances):
     with contraction OFF the text's outputs equal the oracle's convention C bit for bit;
  2. what real compilers do to `dx*dx + dy*dy + dz*dz` in THIS text: g++ and clang++ (LLVM, the family nvcc's NVVM derives from) with
     contraction on both produce the oracle's convention B - never A - which is why B is the default.

One edit is applied to the extracted text: a `__syncthreads();` after `old = dists_i[0];` (sampling_gpu.cu:205).  The kernel has a
benign read/write race there (a fast thread of iteration j+1 may overwrite dists_i[0] before a slow thread of iteration j has read
it) that lock-step warps do not lose; a sequential scheduler always would.
"""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import pn2_oracle as ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/networks/pts_encoder/pointnet2_utils/pointnet2/src"
EMU = os.path.join(ROOT, "oracle", "cu_emul")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


pytestmark = pytest.mark.skipif(not (os.path.isdir(SRC) and shutil.which("g++") and _has_fma()),
                                reason="needs /root/reference, g++ and an FMA-capable host (build container only)")


def _extract(path, start_marker, end_marker):
    """Lines from the one that starts with start_marker up to (not including) the one that starts with end_marker."""
    lines = open(path).read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith(start_marker))
    b = next(i for i, l in enumerate(lines) if i > a and l.startswith(end_marker))
    return "\n".join(lines[a:b]) + "\n"


@pytest.fixture(scope="module")
def inc_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("cu_text")
    fps = _extract(os.path.join(SRC, "sampling_gpu.cu"), "__device__ void __update", "void furthest_point_sampling_kernel_launcher")
    race = "old = dists_i[0];"
    assert fps.count(race) == 1
    fps = fps.replace(race, race + "\n    __syncthreads(); /* test-time edit, see the module docstring */")
    (d / "fps_kernel.inc").write_text(fps)
    (d / "ball_query_kernel.inc").write_text(
        _extract(os.path.join(SRC, "ball_query_gpu.cu"), "__global__ void ball_query_kernel_fast", "void ball_query_kernel_launcher_fast"))
    (d / "three_nn_kernel.inc").write_text(
        _extract(os.path.join(SRC, "interpolate_gpu.cu"), "__global__ void three_nn_kernel_fast", "void three_nn_kernel_launcher_fast"))
    (d / "three_interpolate_kernel.inc").write_text(
        _extract(os.path.join(SRC, "interpolate_gpu.cu"), "__global__ void three_interpolate_kernel_fast",
                 "void three_interpolate_kernel_launcher_fast"))
    return d


def _build(inc_dir, tag, cxx, flags):
    so = inc_dir / f"libcu_emul_{tag}.so"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math", "-w", f"-I{inc_dir}", f"-I{EMU}",
           f'-DEMU_CUDA_UTILS_H="{SRC}/cuda_utils.h"'] + flags + ["-o", str(so), os.path.join(EMU, "driver.cpp")]
    subprocess.check_call(cmd)
    return ctypes.CDLL(str(so))


VARIANTS = {
    # tag: (compiler, flags, the oracle convention the text must reproduce)
    "gxx_off": ("g++", ["-ffp-contract=off"], "C"),
    "gxx_fma": ("g++", ["-mfma", "-ffp-contract=fast"], "B"),
    "clang_fma": (CLANG, ["-mfma", "-ffp-contract=fast"], "B"),
    "clang_default": (CLANG, ["-mfma"], "B"),  # clang's default -ffp-contract=on (front-end fmuladd): the same fusion
}


@pytest.fixture(scope="module", params=list(VARIANTS))
def emu(request, inc_dir):
    cxx, flags, conv = VARIANTS[request.param]
    if not (shutil.which(cxx) or os.path.exists(cxx)):
        pytest.skip(f"{cxx} not present")
    return _build(inc_dir, request.param, cxx, flags), conv


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def emu_fps(lib, xyz, m):
    B, N, _ = xyz.shape
    temp = np.full((B, N), 1e10, np.float32)
    idx = np.zeros((B, m), np.int32)
    lib.emu_fps(B, N, m, _p(xyz), _p(temp), _p(idx))
    return idx, temp


def test_kernel_text_equals_the_oracle(emu, golden):
    lib, conv = emu
    clouds = golden("g1_g2_ops.npz")["clouds"]  # REAL275-shaped surfaces, a tiled-duplicate cloud, an exact-grid-tie cloud
    odd = np.ascontiguousarray(golden("g1_g2_ops.npz")["odd_clouds"])
    assert lib.emu_opt_n_threads(700) == ops.opt_n_threads(700) == 512
    with ops.use_arith(conv):
        cur = clouds
        for lvl, (npnt, radii) in enumerate(zip([512, 256, 128], [[0.02, 0.04], [0.04, 0.08], [0.08, 0.16]])):
            got, got_t = emu_fps(lib, cur, npnt)
            ref, ref_t = ops.furthest_point_sampling(cur, npnt)
            assert np.array_equal(got, ref), f"FPS level {lvl} under {conv}"
            assert np.array_equal(got_t, ref_t)
            new = np.ascontiguousarray(np.take_along_axis(cur, ref[:, :, None].astype(np.int64), axis=1))
            for r, ns in zip(radii, [16, 32]):
                idx = np.zeros((cur.shape[0], npnt, ns), np.int32)
                lib.emu_ball_query(cur.shape[0], cur.shape[1], npnt, ctypes.c_float(r), ns, _p(new), _p(cur), _p(idx))
                assert np.array_equal(idx, ops.ball_query(r, ns, cur, new))
            cur = new
        # n not a power of two: threads own several points (k = tid, tid + S, ...), S = 512 < n = 700
        got, got_t = emu_fps(lib, odd, 60)
        ref, ref_t = ops.furthest_point_sampling(odd, 60)
        assert np.array_equal(got, ref) and np.array_equal(got_t, ref_t)
        # three_nn (double best-distances, strict <) and three_interpolate
        unk, kn = np.ascontiguousarray(clouds[:2, :300]), np.ascontiguousarray(clouds[:2, 300:364])
        d2 = np.zeros((2, 300, 3), np.float32)
        i3 = np.zeros((2, 300, 3), np.int32)
        lib.emu_three_nn(2, 300, 64, _p(unk), _p(kn), _p(d2), _p(i3))
        rd, ri = ops.three_nn(unk, kn)
        assert np.array_equal(i3, ri) and np.array_equal(d2, rd)
        feats = np.random.default_rng(5).standard_normal((2, 7, 64)).astype(np.float32)
        w = np.random.default_rng(6).uniform(size=(2, 300, 3)).astype(np.float32)
        out = np.zeros((2, 7, 300), np.float32)
        lib.emu_three_interpolate(2, 7, 64, 300, _p(feats), _p(ri), _p(w), _p(out))
        assert np.array_equal(out, ops.three_interpolate(feats, ri, w))


def test_no_host_compiler_produces_convention_A(inc_dir, golden):
    """The round-1..4 default (A: the FIRST product rounded on its own) is what neither g++ nor clang++ make of the text: on the golden
    clouds the contracted text picks B's centres, and those differ from A's."""
    clouds = golden("g1_g2_ops.npz")["clouds"]
    a = ops.furthest_point_sampling(clouds, 512, arith="A")[0]
    b = ops.furthest_point_sampling(clouds, 512, arith="B")[0]
    assert not np.array_equal(a, b)
    for tag in ("gxx_fma", "clang_fma"):
        cxx, flags, _ = VARIANTS[tag]
        if not (shutil.which(cxx) or os.path.exists(cxx)):
            continue
        got, _ = emu_fps(_build(inc_dir, tag + "_a", cxx, flags), clouds, 512)
        assert np.array_equal(got, b) and not np.array_equal(got, a)
