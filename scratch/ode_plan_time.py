"""The RK45 driver's launch plans at scripts/eval_single.sh's batch shape (256 clouds x 50 candidates = 12 800 coupled rows, T0 = 0.55):
solve time and time per attempt under the shared-chunk plan (48-row own tiles + left-over chunks shared across stages, one launch per
attempt) against whole tiles (64 / 32 rows, six stage launches per attempt).   python scratch/ode_plan_time.py [B] [K]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genpose_amd.samplers import ODESampler  # noqa: E402
from genpose_amd.scorenet import ScoreNetHIP  # noqa: E402
from genpose_amd.weights_synth import make_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
gen = torch.Generator().manual_seed(1)
cvec, centre, x0 = torch.randn(B, 768, generator=gen).cuda(), torch.randn(B, 3, generator=gen).cuda(), torch.randn(B * K, 9, generator=gen).cuda() * 0.5
FLOP_ROW = 0.5335e6
print(f"{B} clouds x {K} candidates = {B * K} rows, T0 = 0.55, {torch.cuda.get_device_properties(0).multi_processor_count} CUs")
for tile in (0, 64, 32):
    try:
        smp = ODESampler(net, B, K, "cuda", tile=tile)
    except ValueError as e:
        print(f"tile {tile}: {e}")
        continue
    for _ in range(3):
        smp.run(cvec, centre, x0, T0=0.55)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        smp.run(cvec, centre, x0, T0=0.55)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    st = smp.last_stats
    att, nfev = int(st["n_attempts"]), int(st["nfev"])
    dt = statistics.median(ts)
    # the attempts alone: replay the steady-state graph of `att` attempts on a finished state is not meaningful - time whole solves and
    # attribute (nfev - 3) evaluations to the attempts
    tf = B * K * nfev * FLOP_ROW / dt / 1e12
    name = f"plan {smp.plan:#x} (" + ("shared-chunk, " if smp.shared else "") + f"{smp.tile}-row tiles)"
    print(f"{name:44s} solve {dt * 1e3:7.3f} ms (min {min(ts) * 1e3:.3f} max {max(ts) * 1e3:.3f})  attempts {att} nfev {nfev}  "
          f"{dt / att * 1e6:7.1f} us per attempt incl. controller  {tf:6.1f} TF = {tf / 157.3:.3f} of the fp32 MFMA peak")
