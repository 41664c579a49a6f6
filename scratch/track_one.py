"""One tracking sequence, frame by frame (for rocprofv3 --kernel-trace --stats: kernel time per frame against the wall time per frame)."""
import sys, time; sys.path.insert(0, '.')
import torch
from genpose_amd import synth
from genpose_amd.config import get_config
from genpose_amd.posenet_agent import PoseNet
from genpose_amd.runner import TrackingRunner
from genpose_amd.weights_synth import make_state_dict
sa = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"])); sa.load_state_dict(make_state_dict(0, "score"))
ea = PoseNet(get_config(posenet_mode="energy")); ea.load_state_dict(make_state_dict(0, "energy"))
n_obj, K, nfr, warm = 5, 50, 106, 6
base = torch.from_numpy(synth.make_batch(n_obj, start=0))
gt = torch.eye(4).repeat(n_obj, 1, 1); gt[:, :3, 3] = base.mean(dim=1)
frames = [(base + 0.002 * (f % 30)).cuda() for f in range(nfr)]
names = [f"o{j}" for j in range(n_obj)]
tr = TrackingRunner(sa, ea, repeat_num=K, T0=0.15)
for f in range(warm): tr.step(frames[f], names, gt)
torch.cuda.synchronize(); t = time.time()
for f in range(warm, nfr): tr.step(frames[f], names, gt)
torch.cuda.synchronize(); dt = time.time() - t
print(f"frames {nfr - warm}: {dt / (nfr - warm) * 1e3:.3f} ms per frame; replays {sa.net.last_sampler.last_replays}")
