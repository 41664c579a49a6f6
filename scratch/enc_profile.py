"""Encoder alone (tuning): python scratch/enc_profile.py B [iters] [mode] - HIP-event time per pass; run under rocprofv3 --kernel-trace --stats for
the per-kernel table.  mode: forward (everything on one stream, launch by launch) | pass (the agent path's pass: deeper sampling levels on a side
stream under level 0, launch by launch) | graph (the same pass as one hipGraph replay)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genpose_amd import synth
from genpose_amd.encoder import Pointnet2EncoderHIP
from genpose_amd.weights_synth import make_state_dict
B = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "forward"
prec = sys.argv[4] if len(sys.argv) > 4 else os.environ.get("GP_ENC_PRECISION", "f32")
enc = Pointnet2EncoderHIP(make_state_dict(0, "score"), "cuda", precision=prec)
pts = torch.from_numpy(synth.make_batch(B)).cuda()
run = {"forward": lambda: enc.forward(pts), "pass": lambda: enc.encode(pts, use_graph=False)[0], "graph": lambda: enc.encode(pts)[0]}[mode]
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / iters
print(f"encoder B={B} [{mode}, {prec}]: {t:.3f} ms per pass = {B * 2.201 / t:.1f} TFLOP/s on the reference count")
