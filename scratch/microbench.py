"""Per-entry-point timing on the GPU (HIP events), for tuning.  Usage: python scratch/microbench.py [B] [K]"""
import sys, os; sys.path.insert(0, '.')
import torch, numpy as np
from genpose_amd import synth
from genpose_amd.encoder import Pointnet2EncoderHIP
from genpose_amd.scorenet import ScoreNetHIP
from genpose_amd.samplers import PCSampler
from genpose_amd.weights_synth import make_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sd = make_state_dict(0, "score")
enc = Pointnet2EncoderHIP(sd, "cuda", precision=os.environ.get("GP_ENC_PRECISION", "f32")); net = ScoreNetHIP(sd, "cuda")
pts = torch.from_numpy(synth.make_batch(B)).cuda()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_enc = timeit(lambda: enc.forward(pts))
feat = enc.forward(pts); cvec = net.cloud_embed(feat)
smp = PCSampler(net, B, K, 100, "cuda", use_graph=True, precision=os.environ.get("GP_SMP_PRECISION", "f32"))
x0 = torch.randn(B * K, 9, device="cuda") * 50
t_pc = timeit(lambda: smp.run(cvec, pts.mean(1), x0), 5)
print(f"B={B} K={K} ({smp.kernel_name}): encoder {t_enc:.3f} ms ({B*2.201/t_enc:.1f} TFLOP/s), "
      f"PC-100 {t_pc:.3f} ms ({t_pc*10:.1f} us/step, {B*K*0.5335e-3*101/t_pc:.1f} TFLOP/s)")
