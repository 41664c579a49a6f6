import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, numpy as np
from oracle import genpose_oracle as go
from genpose_amd.config import get_config
from genpose_amd.posenet_agent import PoseNet
g = np.load('tests/golden/g6_ode.npz'); case = 'T1_none'
agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"], sampling_steps=None)); agent.load_state_dict(go.make_state_dict(0, "score"))
pts = torch.from_numpy(g["pts"]).cuda()
prior = torch.from_numpy(g[f"{case}_prior_noise"])
agent.net.prior_fn = lambda shape, T=1.0: prior * float(go.ve_sigma(torch.tensor(T)))
out = agent.pred_func({"pts": pts, "pts_center": pts.mean(dim=1)}, repeat_num=10, save_path=None, T0=1.0)
st = agent.net._samplers[("ode", 2, 10)].last_stats
log = []
ref, _, nfev = go.pred_func(go.make_state_dict(0, "score"), pts.cpu(), pts.cpu().mean(dim=1), 10, "ode", prior, T0=1.0, log=log)
print("nfev", st["nfev"], nfev, "maxdiff", float((out.cpu() - ref).abs().max()), "scale", float(ref.abs().max()))
for i in range(min(len(log), len(st["log_err"]))):
    print(i, f"t {log[i]['t']:.6f} {st['log_t'][i]:.6f}  h {log[i]['h']:.3e} {st['log_h'][i]:.3e}  err {log[i]['err_norm']:.4f} {st['log_err'][i]:.4f}  acc {int(log[i]['accepted'])} {st['log_acc'][i]}")
