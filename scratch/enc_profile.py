"""Encoder alone (tuning): python scratch/enc_profile.py B [iters] - HIP-event time per pass; run under rocprofv3 --kernel-trace --stats for the per-kernel table."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genpose_amd import synth
from genpose_amd.encoder import Pointnet2EncoderHIP
from genpose_amd.weights_synth import make_state_dict
B = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
enc = Pointnet2EncoderHIP(make_state_dict(0, "score"), "cuda")
pts = torch.from_numpy(synth.make_batch(B)).cuda()
for _ in range(3): enc.forward(pts)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): enc.forward(pts)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / iters
print(f"encoder B={B}: {t:.3f} ms per pass = {B * 2.201 / t:.1f} TFLOP/s on the reference count")
