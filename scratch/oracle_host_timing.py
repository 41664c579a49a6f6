"""Host-side timing of the CPU oracle pieces the full-size GPU tests wait for (run on the GPU box; no device work)."""
import os, time, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genpose_amd import synth
from oracle import genpose_oracle as go, parallel

if __name__ == "__main__":
    print("cpu_count", os.cpu_count(), "torch threads default", torch.get_num_threads())
    sd = go.make_state_dict(0, "score")
    p = synth.make_batch(128, start=7000)
    t = time.time(); parallel.encoder_features("score", p); print(f"pool encoder 128 clouds: {time.time() - t:.1f} s (plan {parallel._plan()})")
    t = time.time(); parallel.encoder_features("score", synth.make_batch(256, start=9000)); print(f"pool encoder 256 more clouds (warm pool): {time.time() - t:.1f} s")
    for nt in (256, 8, 32):
        torch.set_num_threads(nt)
        t = time.time(); go.encoder_forward(sd, torch.from_numpy(p[:32])); print(f"direct encoder 32 clouds, {nt} threads: {time.time() - t:.1f} s")
    R = 12800
    feat = torch.randn(R, 1024); x = torch.randn(R, 9); tt = torch.full((R, 1), 0.5)
    for nt in (256, 128, 64, 32, 16):
        torch.set_num_threads(nt)
        go.score_forward(sd, feat, x, tt)
        t = time.time()
        for _ in range(5):
            go.score_forward(sd, feat, x, tt)
        print(f"score_forward {R} rows, {nt} threads: {(time.time() - t) / 5 * 1e3:.0f} ms per evaluation")
