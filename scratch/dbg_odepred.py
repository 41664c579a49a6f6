import sys, time; sys.path.insert(0, '.')
import torch
from genpose_amd import synth
from genpose_amd.config import get_config
from genpose_amd.posenet_agent import PoseNet
from genpose_amd.pipeline import GroupedODEPredictor
from genpose_amd.weights_synth import make_state_dict
agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"])); agent.load_state_dict(make_state_dict(0, "score"))
pts = torch.from_numpy(synth.make_batch(64)).cuda()
for G in (1, 5):
    pred = GroupedODEPredictor(agent, 64, 50, T0=0.55, batches_per_launch=G)
    pred.run([pts] * G); torch.cuda.synchronize()
    t = time.time(); pred.run([pts] * 10); torch.cuda.synchronize(); dt = time.time() - t
    print(f"G={G}: {dt/10*1e3:.2f} ms per batch, nfev {pred.last_nfev}")
    # breakdown of one group
    grp = torch.cat([pts] * G) if G > 1 else pts
    torch.cuda.synchronize(); t = time.time(); f = agent.net.pts_encoder(grp); c = agent.net.pose_score_net.cloud_embed(f); torch.cuda.synchronize(); t_enc = time.time() - t
    t = time.time(); x0 = agent.net.prior_fn((G * 3200, 9), T=0.55).cuda(); torch.cuda.synchronize(); t_prior = time.time() - t
    smp = pred._sampler(G)
    t = time.time(); smp.run(c, grp.mean(1), x0, 0.55); torch.cuda.synchronize(); t_ode = time.time() - t
    print(f"   encoder {t_enc*1e3:.2f} ms, prior {t_prior*1e3:.2f} ms, ode {t_ode*1e3:.2f} ms")
