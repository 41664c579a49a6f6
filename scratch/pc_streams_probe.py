"""One 64-cloud batch per launch chain (the literal configs[1] shape: 3 200 rows, 16-row tiles, 200 of 256 CUs) with SEVERAL chains in flight:
PipelinedPCPredictor(sampler_streams = s, depth = d), batches_per_launch = 1.  Round 1 measured two chains slower than one; re-measured on the
round-6 kernels.    python scratch/pc_streams_probe.py"""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genpose_amd import synth  # noqa: E402
from genpose_amd.config import get_config  # noqa: E402
from genpose_amd.pipeline import PipelinedPCPredictor  # noqa: E402
from genpose_amd.posenet_agent import PoseNet  # noqa: E402
from genpose_amd.weights_synth import make_state_dict  # noqa: E402

B, K, n = 64, 50, 100
agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=n))
agent.load_state_dict(make_state_dict(0, "score"))
pool = [torch.from_numpy(synth.make_batch(B, start=B * j)).cuda() for j in range(12)]
ref = None
for s, d, G in ((1, 2, 1), (2, 2, 1), (2, 4, 1), (3, 3, 1), (3, 6, 1), (4, 4, 1), (1, 2, 2), (2, 4, 2), (1, 2, 10)):
    pipe = PipelinedPCPredictor(agent, B, K, n, depth=d, sampler_streams=s, batches_per_launch=G)
    nb = 48
    batches = [pool[j % len(pool)] for j in range(nb)]
    pipe.run(batches[: 2 * max(s, G)])
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        pipe.run(batches)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    print(f"batches per launch {G:2d}, sampler chains in flight {s}, slots {d}: {B * nb / dt:9.0f} poses/s  ({dt / nb * 1e3:.3f} ms per batch; min {min(ts) / nb * 1e3:.3f} max {max(ts) / nb * 1e3:.3f})", flush=True)
