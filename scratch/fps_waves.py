"""Furthest point sampling (one wave per cloud for n <= 1024): python scratch/fps_waves.py

(Round 5 compared one / two / four waves per cloud through a tuning-only switch: profiles/r5_fps_one_wave.txt.)  The kernel is first held to the oracle (indices of every level, bit for bit, all three distance conventions), then timed with HIP events:
the encoder's chain (1024 -> 512 -> 128), level 0 alone, and the stand-alone operator at a few (n, m).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genpose_amd import _lib, synth  # noqa: E402
from genpose_amd._lib import ptr, stream_ptr  # noqa: E402
from genpose_amd.config import dist_arith_code  # noqa: E402
from oracle import pn2_oracle  # noqa: E402

import ctypes  # noqa: E402


def chain(xyz, ms, arith):
    B, N, _ = xyz.shape
    idx, new = [], []
    prev = N
    for m in ms:
        idx.append(torch.empty(B, m, dtype=torch.int32, device=xyz.device))
        new.append(torch.empty(B, m, 3, dtype=torch.float32, device=xyz.device))
        prev = m
    arr = (ctypes.c_int * 3)(*(list(ms) + [0] * (3 - len(ms))))
    pi = [ptr(t) for t in idx] + [None] * (3 - len(ms))
    px = [ptr(t) for t in new] + [None] * (3 - len(ms))
    _lib.call("gp_fps_chain_arith", dist_arith_code(arith), B, N, len(ms), arr, ptr(xyz), pi[0], px[0], pi[1], px[1], pi[2], px[2], stream_ptr())
    return idx, new


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    pts = torch.from_numpy(synth.make_batch(64, start=5)).to(dev).contiguous()
    # parity first
    for arith in "ABC":
        for ms in ((512, 128), (512, 256, 128), (300, 77)):
            idx, new = chain(pts[:8], ms, arith)
            cur = pts[:8].cpu().numpy()
            for l, m in enumerate(ms):
                ref = pn2_oracle.furthest_point_sampling(cur, m, arith=arith)[0]
                got = idx[l].cpu().numpy()
                assert np.array_equal(ref, got), (arith, ms, l)
                cur = np.take_along_axis(cur, ref[..., None].astype(np.int64), axis=1)
                assert np.array_equal(cur, new[l].cpu().numpy())
    for n, m in ((1024, 512), (700, 60), (64, 64), (100, 37), (4096, 512), (1500, 40)):
        x = torch.randn(3, n, 3, device=dev)
        if n >= 200:
            x[:, 100:200] = x[:, 0:100]  # exact duplicates: ties
        temp = torch.full((3, n), 1e10, device=dev)
        idx = torch.empty(3, m, dtype=torch.int32, device=dev)
        _lib.call("gp_furthest_point_sampling", 3, n, m, ptr(x), ptr(temp), ptr(idx), stream_ptr())
        ref, rtemp = pn2_oracle.furthest_point_sampling(x.cpu().numpy(), m)
        assert np.array_equal(ref, idx.cpu().numpy()), (n, m)
        assert np.array_equal(rtemp, temp.cpu().numpy()), (n, m)
    print(f"parity with the oracle: chain (3 conventions, 3 level sets), operator at 6 shapes incl. ties - OK")
    for B in (5, 64, 320, 640):
        x = torch.from_numpy(synth.make_batch(B)).to(dev).contiguous()
        t_chain = timed(lambda: chain(x, (512, 256, 128), "B"))
        t_l0 = timed(lambda: chain(x, (512,), "B"))
        print(f"{B:4d} clouds: chain 1024->512->256->128 {t_chain:7.1f} us   level 0 alone {t_l0:7.1f} us")


if __name__ == "__main__":
    main()
