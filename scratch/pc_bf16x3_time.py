"""PC step launch time: fp32 plans vs the opt-in split-bf16 trunk (python scratch/pc_bf16x3_time.py)."""
import sys; sys.path.insert(0, ".")
import torch
from genpose_amd.samplers import PCSampler
from genpose_amd.scorenet import ScoreNetHIP
from genpose_amd.weights_synth import make_state_dict
net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
B1, K, n = 64, 50, 40
for G in (1, 2, 5, 10, 20, 40):
    row = []
    for prec in ("f32", "bf16x3"):
        smp = PCSampler(net, G * B1, K, n, "cuda", groups=G, precision=prec)
        cvec, cen, x0 = torch.randn(G * B1, 768, device="cuda"), torch.randn(G * B1, 3, device="cuda"), torch.randn(G * B1 * K, 9, device="cuda") * 50
        smp.run(cvec, cen, x0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): smp.graph.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 / (n + 1) * 1e3
        row.append((smp.kernel_name, us))
    R = G * B1 * K
    tf = lambda us: R * 0.5335e6 / us / 1e6
    print(f"{R:6d} rows: {row[0][0]:26s} {row[0][1]:7.1f} us ({tf(row[0][1]):6.1f} TFLOP/s)   {row[1][0]} {row[1][1]:7.1f} us ({tf(row[1][1]):6.1f} TFLOP/s of fp32-equivalent work)   {row[0][1] / row[1][1]:.2f}x")
