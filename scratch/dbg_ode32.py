import sys; sys.path.insert(0, '.')
import torch, numpy as np
from oracle import genpose_oracle as go
from genpose_amd.scorenet import ScoreNetHIP
from genpose_amd.samplers import ODESampler
sd = go.make_state_dict(0, "score")
snet = ScoreNetHIP(sd, "cuda")
B, K, T0 = 90, 50, 0.3
gen = torch.Generator().manual_seed(9)
pf = torch.randn(B, 1024, generator=gen).abs()
centre = torch.randn(B, 3, generator=gen) * 0.3
init_x = torch.randn(B * K, 9, generator=gen) * float(go.ve_sigma(torch.tensor(T0)))
log = []
fr = pf.repeat_interleave(K, 0)
_, ref, nfev = go.ode_sampler(lambda x, t: go.score_forward(sd, fr, x, t), init_x, centre.repeat_interleave(K, 0), T0, log=log)
smp = ODESampler(snet, B, K, "cuda")
_, x = smp.run(snet.cloud_embed(pf.cuda()), centre.cuda(), init_x.cuda(), T0)
st = smp.last_stats
print("tile", smp.tile if hasattr(smp, "tile") else None, "nfev", st["nfev"], nfev, "attempts", len(st["log_err"]), len(log))
for i in range(max(len(log), len(st["log_err"]))):
    a = log[i] if i < len(log) else None
    b = (st["log_t"][i], st["log_h"][i], st["log_err"][i], st["log_acc"][i]) if i < len(st["log_err"]) else None
    print(i, a, b)
print("maxdiff", float((x.cpu() - ref).abs().max()))
