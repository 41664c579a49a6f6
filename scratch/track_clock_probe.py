"""Is the slow-down of a lightly loaded tracking loop (fast for ~0.2 s, then 30 % slower) the chip's clock management?  Per repeat of 60 frames:
ms per frame and the shader clock rocm-smi reports right after it; then the same with a 30 ms burst of dense work before every repeat."""
import os
import re
import statistics
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genpose_amd import runner, synth  # noqa: E402
from genpose_amd.config import get_config  # noqa: E402
from genpose_amd.posenet_agent import PoseNet  # noqa: E402
from genpose_amd.weights_synth import make_state_dict  # noqa: E402


def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        m = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        return m[0] if m else out.strip().splitlines()[-3:]
    except Exception as e:  # noqa: BLE001
        return repr(e)


sa = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"])); sa.load_state_dict(make_state_dict(0, "score"))
ea = PoseNet(get_config(posenet_mode="energy")); ea.load_state_dict(make_state_dict(0, "energy"))
n_obj, K, nfr = 5, 50, 30
base = torch.from_numpy(synth.make_batch(n_obj, start=0))
gt = torch.eye(4).repeat(n_obj, 1, 1); gt[:, :3, 3] = base.mean(dim=1)
frames = [(base + 0.002 * f).cuda() for f in range(nfr)]
names = [f"o{j}" for j in range(n_obj)]
big = torch.randn(8192, 8192, device="cuda")
print("idle sclk:", sclk())
for burst in (False, True, False):
    tr = runner.TrackingRunner(sa, ea, repeat_num=K, T0=0.15)
    for f in range(8):
        tr.step(frames[f % nfr], names, gt)
    torch.cuda.synchronize()
    per, clk = [], []
    for rep in range(8):
        if burst:
            for _ in range(6):
                big @ big
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(60):
            tr.step(frames[(8 + rep * 60 + f) % nfr], names, gt)
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / 60 * 1e3)
        clk.append(sclk() if not burst else "-")
    print(f"burst before every repeat: {burst}  median {statistics.median(per):.3f}  by repeat {[round(p, 3) for p in per]}  sclk after each {clk}", flush=True)
# the gap between repeats matters?  (the rocm-smi call above idles the GPU for ~0.1 s between repeats)
tr = runner.TrackingRunner(sa, ea, repeat_num=K, T0=0.15)
for f in range(8):
    tr.step(frames[f % nfr], names, gt)
torch.cuda.synchronize()
per = []
for rep in range(12):
    t0 = time.perf_counter()
    for f in range(60):
        tr.step(frames[(8 + rep * 60 + f) % nfr], names, gt)
    torch.cuda.synchronize()
    per.append((time.perf_counter() - t0) / 60 * 1e3)
print(f"back to back, 12 repeats: {[round(p, 3) for p in per]}")
