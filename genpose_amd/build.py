"""Builds genpose_amd/lib/libgenpose_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m genpose_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# GP_BUILD_TAG=<tag> (tuning): build a variant beside the shipped library (lib/libgenpose_hip_<tag>.so, objects in lib/obj_<tag>/);
# select it at run time with GENPOSE_HIP_LIB=<path>
TAG = os.environ.get("GP_BUILD_TAG", "")
SO = os.path.join(LIBDIR, f"libgenpose_hip_{TAG}.so" if TAG else "libgenpose_hip.so")
OBJDIR = os.path.join(LIBDIR, f"obj_{TAG}") if TAG else LIBDIR
SOURCES = ["misc.hip", "pn2_ops.hip", "sa_mlp.hip", "scorenet.hip", "rk45.hip", "rank.hip", "preprocess.hip", "score_div.hip", "sa_bf16x3.hip", "trunk_bf16x3.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]
FLAGS += [f for f in os.environ.get("GP_EXTRA_FLAGS", "").split() if f]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    d.append(os.path.join(os.path.dirname(HERE), "include", "genpose_hip.h"))
    return d


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    newest = max(os.path.getmtime(p) for p in _deps())
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= newest:
        return SO
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [_hipcc()] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
