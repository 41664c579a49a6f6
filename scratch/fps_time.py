"""FPS chain alone (tuning): python scratch/fps_time.py B ... ; GP_FPS_WAVE_MAXN selects the levels that run on one wave."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genpose_amd import synth
from genpose_amd.encoder import Pointnet2EncoderHIP
from genpose_amd.weights_synth import make_state_dict
enc = Pointnet2EncoderHIP(make_state_dict(0, "score"), "cuda")
for B in [int(a) for a in sys.argv[1:]] or [64, 320]:
    pts = torch.from_numpy(synth.make_batch(B)).cuda()
    for _ in range(3): enc.sample_centres(pts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): enc.sample_centres(pts)
    e1.record(); torch.cuda.synchronize()
    print(f"GP_FPS_WAVE_MAXN={os.environ.get('GP_FPS_WAVE_MAXN')}: B={B}: fps chain {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
