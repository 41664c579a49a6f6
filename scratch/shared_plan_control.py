"""Control for shared_plan_soak.py: on the same 100 inputs, how far apart are two WHOLE-tile plans (64- and 32-row tiles), and where do the
largest differences of the shared-chunk plan sit (own rows or shared-chunk rows)?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genpose_amd.samplers import ODESampler  # noqa: E402
from genpose_amd.scorenet import ScoreNetHIP  # noqa: E402
from genpose_amd.weights_synth import make_state_dict  # noqa: E402
net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
B, K = 256, 50
s, t64, t32 = ODESampler(net, B, K, "cuda"), ODESampler(net, B, K, "cuda", tile=64), ODESampler(net, B, K, "cuda", tile=32)
w = {"shared-64": [], "64-32": [], "shared-32": []}
where = []
for seed in range(100):
    gen = torch.Generator().manual_seed(1000 + seed)
    cvec, centre = torch.randn(B, 768, generator=gen).cuda(), torch.randn(B, 3, generator=gen).cuda()
    x0 = (torch.randn(B * K, 9, generator=gen) * (0.3 + 0.1 * (seed % 7))).cuda()
    xs, x64, x32 = (m.run(cvec, centre, x0, T0=0.55)[1].clone() for m in (s, t64, t32))
    d = (xs - x64)[:, :6].abs()
    w["shared-64"].append(float(d.max())); w["64-32"].append(float((x64 - x32)[:, :6].abs().max())); w["shared-32"].append(float((xs - x32)[:, :6].abs().max()))
    r = int(d.max(dim=1).values.argmax())
    where.append((seed, r, r >= 12288, float(d.max()), float(d[12288:].max()), float(torch.quantile(d.max(dim=1).values, 0.999))))
for k, v in w.items():
    v = np.array(v)
    print(f"{k:10s} rotation block, max abs difference per solve: median {np.median(v):.2e}  p90 {np.quantile(v, 0.9):.2e}  max {v.max():.2e}")
bad = sorted(where, key=lambda t: -t[3])[:6]
for seed, r, inshared, dmax, dshared, p999 in bad:
    print(f"seed {seed}: worst row {r} ({'shared chunk' if inshared else 'own tile'}) {dmax:.2e}; worst over the shared-chunk rows {dshared:.2e}; p99.9 over rows {p999:.2e}")
print("solves whose worst row lies in a shared chunk:", sum(1 for t in where if t[2]), "of", len(where), "(512 of 12 800 rows = 4 % are shared-chunk rows)")
