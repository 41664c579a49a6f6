"""Sampler launch chain alone (tuning): python scratch/pc_time.py B [B ...] - HIP-event time per pc_step launch (graph replay)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genpose_amd.scorenet import ScoreNetHIP
from genpose_amd.samplers import PCSampler
from genpose_amd.weights_synth import make_state_dict
K, n = 50, 100
net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
for B in [int(a) for a in sys.argv[1:]] or [64, 320]:
    G = B // 64 if B % 64 == 0 else 1
    cvec = torch.randn(B, 768, device="cuda"); cen = torch.zeros(B, 3, device="cuda")
    x0 = torch.randn(B * K, 9, device="cuda") * 50
    smp = PCSampler(net, B, K, n, "cuda", use_graph=True, groups=G)
    for _ in range(2): smp.run(cvec, cen, x0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): smp.graph.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps / (n + 1)
    print(f"{os.path.basename(os.environ.get('GENPOSE_HIP_LIB', 'default'))}: rows {B*K} tile {smp.tile} groups {G}: {us:.2f} us per launch, "
          f"{B*K*0.5335e-3/us:.1f} TFLOP/s = {B*K*0.5335e-3/us/157.3:.3f} of peak", flush=True)
