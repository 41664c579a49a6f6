"""Where a PCSampler.run() call goes beside its kernels (tuning): python scratch/pc_run_pieces.py [B] [groups]"""
import sys; sys.path.insert(0, '.')
import torch
from genpose_amd.scorenet import ScoreNetHIP
from genpose_amd.samplers import PCSampler
from genpose_amd.weights_synth import make_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 640
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10
K = 50
net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
smp = PCSampler(net, B, K, 100, "cuda", use_graph=True, groups=G)
cvec = torch.randn(B, 768, device="cuda"); cen = torch.randn(B, 3, device="cuda"); x0 = torch.randn(B * K, 9, device="cuda") * 50
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
smp.run(cvec, cen, x0); smp.run(cvec, cen, x0); torch.cuda.synchronize()
print(f"run()            {timeit(lambda: smp.run(cvec, cen, x0)):9.1f} us")
print(f"graph.replay()   {timeit(lambda: smp.graph.replay()):9.1f} us")
print(f"2 x normal_      {timeit(lambda: (smp.z1.normal_(), smp.z2.normal_())):9.1f} us")
print(f"3 input copies   {timeit(lambda: (smp.cvec.copy_(cvec), smp.centre.copy_(cen), smp.x.copy_(x0))):9.1f} us")
