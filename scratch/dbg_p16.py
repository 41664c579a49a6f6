import sys, os; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import genpose_oracle as go
from genpose_amd.scorenet import ScoreNetHIP
sd = go.make_state_dict(0, "score")
snet = ScoreNetHIP(sd, "cuda")
gen = torch.Generator().manual_seed(3)
for B, K in [(2, 10), (3, 50), (64, 50)]:
    pf = torch.randn(B, 1024, generator=gen).abs()
    pose = torch.randn(B * K, 9, generator=gen)
    t = 0.3
    ref = go.score_forward(sd, pf.repeat_interleave(K, 0), pose, torch.ones(B * K, 1) * t).numpy()
    cvec = snet.cloud_embed(pf.cuda())
    tvec = snet.time_embed(torch.tensor([t], device="cuda"))
    sigma = torch.tensor([0.01 * 5000.0 ** t], device="cuda")
    got = snet.evaluate(cvec, K, pose.cuda(), tvec[0], sigma, "score").cpu().numpy()
    err = np.abs(got - ref) / (np.abs(ref).max())
    bad = np.argwhere(err > 1e-3)
    print(os.environ.get("GP_SCORE_P"), B, K, "max rel err", err.max(), "bad rows:", sorted(set(bad[:, 0].tolist()))[:20], "cols", sorted(set(bad[:, 1].tolist())))
