"""Tracking throughput (BASELINE configs[4] shape: 5 objects per frame, K = 50, T0 = 0.15 warm start, ODE sampler): frames/s of one
TrackingRunner vs MultiSequenceTracker with S concurrent sequences.  python scratch/bench_tracking.py [S ...]"""
import sys, time; sys.path.insert(0, '.')
import torch
from genpose_amd import synth
from genpose_amd.config import get_config
from genpose_amd.posenet_agent import PoseNet
from genpose_amd.runner import MultiSequenceTracker, TrackingRunner
from genpose_amd.weights_synth import make_state_dict
sa = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"])); sa.load_state_dict(make_state_dict(0, "score"))
ea = PoseNet(get_config(posenet_mode="energy")); ea.load_state_dict(make_state_dict(0, "energy"))
n_obj, K, nfr, warm = 5, 50, 30, 6  # the solver sizes its attempt graph after the first frames: warm up past that
def seq(s):
    base = torch.from_numpy(synth.make_batch(n_obj, start=50 * s))
    gt = torch.eye(4).repeat(n_obj, 1, 1); gt[:, :3, 3] = base.mean(dim=1)
    return [(base + 0.002 * f).cuda() for f in range(nfr)], [f"s{s}o{j}" for j in range(n_obj)], gt
tr = TrackingRunner(sa, ea, repeat_num=K, T0=0.15)
fr, names, gt = seq(0)
for f in range(warm): tr.step(fr[f], names, gt)
torch.cuda.synchronize(); t = time.time()
for f in range(warm, nfr): tr.step(fr[f], names, gt)
torch.cuda.synchronize(); dt = time.time() - t
print(f"single sequence: {(nfr-warm)/dt:.1f} frames/s ({dt/(nfr-warm)*1e3:.2f} ms per frame, {n_obj*(nfr-warm)/dt:.0f} poses/s; attempts per replay {sa.net.last_sampler.last_replays})")
for S in [int(a) for a in sys.argv[1:]] or [8, 32]:
    seqs = [seq(s) for s in range(S)]
    mt = MultiSequenceTracker(sa, ea, S, repeat_num=K, T0=0.15)
    for f in range(warm): mt.step([(q[0][f], q[1], q[2]) for q in seqs])
    torch.cuda.synchronize(); t = time.time()
    for f in range(warm, nfr): r = mt.step([(q[0][f], q[1], q[2]) for q in seqs])
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"{S} sequences per step: {S*(nfr-warm)/dt:.1f} frames/s ({dt/(nfr-warm)*1e3:.2f} ms per step, {S*n_obj*(nfr-warm)/dt:.0f} poses/s, nfev {[x['nfev'] for x in r][:4]}...)")
