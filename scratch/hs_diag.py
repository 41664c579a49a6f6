import sys; sys.path.insert(0, ".")
import torch, numpy as np
from genpose_amd.samplers import ODESampler
from genpose_amd.scorenet import ScoreNetHIP
from genpose_amd.weights_synth import make_state_dict
HS = 0x100
net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
torch.manual_seed(0)
for B, K in ((1, 10), (3, 50), (5, 50)):
    cvec, cen = torch.randn(B, 768, device="cuda"), torch.randn(B, 3, device="cuda")
    x0 = torch.randn(B * K, 9, device="cuda") * 0.04
    for plan in (16, 16 | HS, 16, 16 | HS):
        smp = ODESampler(net, B, K, "cuda", tile=plan)
        _, x = smp.run(cvec, cen, x0, 0.15)
        st = smp.last_stats
        print(B, K, plan, "nfev", st["nfev"], "err", np.array2string(st["log_err"][:8], precision=12), "h", np.array2string(st["log_h"][:4], precision=12), "acc", st["log_acc"][:8])
