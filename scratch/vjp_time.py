"""Launch time of the forward + vector-Jacobian right-hand sides (energy model's score inside a PC step; likelihood / energy-model RK45
stages) at a given row count: python scratch/vjp_time.py [rows=32000] [tile=0]"""
import sys
import torch
sys.path.insert(0, ".")
from genpose_amd.samplers import PCSampler
from genpose_amd.scorenet import ScoreNetHIP
from genpose_amd.weights_synth import make_state_dict
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K = 50
B = rows // K
n = 20
FLOP_ROW = 0.5335e6
for model, flop in (("score", FLOP_ROW), ("energy", 2 * FLOP_ROW)):
    net = ScoreNetHIP(make_state_dict(0, model), "cuda")
    smp = PCSampler(net, B, K, n, "cuda", model=model, tile=tile)
    cvec, centre, x0 = torch.randn(B, 768, device="cuda"), torch.randn(B, 3, device="cuda"), torch.randn(B * K, 9, device="cuda") * 50
    for _ in range(2):
        smp.run(cvec, centre, x0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        smp.graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 3 / n  # n full launches + the short finish launch
    print(f"{model:7s} {smp.kernel_name:28s} rows {B * K}: {us:7.1f} us per launch = {B * K * flop / us / 1e6:6.1f} TFLOP/s = {B * K * flop / us / 1e6 / 157.3:.3f} of the fp32 MFMA peak")
