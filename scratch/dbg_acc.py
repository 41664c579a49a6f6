"""fp32 error of the HIP score trunk vs a float64 evaluation, next to the CPU fp32 oracle's own error."""
import sys; sys.path.insert(0, '.')
import torch, numpy as np
from oracle import genpose_oracle as go
from genpose_amd.scorenet import ScoreNetHIP
sd = go.make_state_dict(0, "score")
sd64 = {k: v.double() for k, v in sd.items()}
snet = ScoreNetHIP(sd, "cuda")
gen = torch.Generator().manual_seed(3)
for B, K in [(2, 10), (64, 50), (128, 50)]:
    pf = torch.randn(B, 1024, generator=gen).abs(); pose = torch.randn(B * K, 9, generator=gen)
    for t in (1e-5, 0.05, 0.5):
        tt = torch.ones(B * K, 1) * t
        ref64 = go.score_forward(sd64, pf.double().repeat_interleave(K, 0), pose.double(), tt.double())
        ref32 = go.score_forward(sd, pf.repeat_interleave(K, 0), pose, tt)
        cvec = snet.cloud_embed(pf.cuda()); tvec = snet.time_embed(torch.tensor([t], device="cuda"))
        sigma = torch.tensor([0.01 * 5000.0 ** t], device="cuda")
        got = snet.evaluate(cvec, K, pose.cuda(), tvec[0], sigma, "score").cpu().double()
        sc = float(ref64.abs().max())
        print(f"B={B} K={K} t={t}: hip-f64 {float((got-ref64).abs().max())/sc:.2e}  cpu32-f64 {float((ref32.double()-ref64).abs().max())/sc:.2e}")
