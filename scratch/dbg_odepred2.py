import sys, time; sys.path.insert(0, '.')
import torch
from genpose_amd import synth
from genpose_amd.config import get_config
from genpose_amd.posenet_agent import PoseNet
from genpose_amd.pipeline import GroupedODEPredictor
from genpose_amd.weights_synth import make_state_dict
agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"])); agent.load_state_dict(make_state_dict(0, "score"))
pts = torch.from_numpy(synth.make_batch(64)).cuda()
pred = GroupedODEPredictor(agent, 64, 50, T0=0.55, batches_per_launch=5)
pred.run([pts] * 5); torch.cuda.synchronize()
smp = pred._sampler(5)
orig = smp.run
def timed(*a, **k):
    torch.cuda.synchronize(); t = time.time(); r = orig(*a, **k); torch.cuda.synchronize()
    print("   ode run %.2f ms, attempts %s" % ((time.time() - t) * 1e3, [int(s["n_attempts"]) for s in smp.group_stats]))
    return r
smp.run = timed
for n in (5, 20):
    t = time.time(); pred.run([pts] * n); torch.cuda.synchronize(); print(f"{n} batches: {(time.time()-t)/n*1e3:.2f} ms per batch")
