import sys, time; sys.path.insert(0, '.')
import torch
from genpose_amd import synth, samplers
from genpose_amd.config import get_config
from genpose_amd.posenet_agent import PoseNet
from genpose_amd.pipeline import GroupedODEPredictor
from genpose_amd.weights_synth import make_state_dict
agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"])); agent.load_state_dict(make_state_dict(0, "score"))
pts = torch.from_numpy(synth.make_batch(64)).cuda()
pred = GroupedODEPredictor(agent, 64, 50, T0=0.55, batches_per_launch=5)
pred.run([pts] * 5); torch.cuda.synchronize()
smp = pred._sampler(5)
orig_read = smp._read_states
log = []
def rd():
    t = time.time(); r = orig_read(); log.append((time.time(), time.time() - t, [int(s["n_attempts"]) for s in r])); return r
smp._read_states = rd
t0 = time.time(); pred.run([pts] * 20); torch.cuda.synchronize(); print("total per batch %.2f ms" % ((time.time() - t0) / 20 * 1e3))
prev = t0
for ts, dt, att in log:
    print("  +%.2f ms (read waited %.2f ms) attempts %s" % ((ts - prev) * 1e3, dt * 1e3, att)); prev = ts
