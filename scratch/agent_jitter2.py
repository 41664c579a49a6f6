import sys, time; sys.path.insert(0, '.')
import torch
from genpose_amd import synth
from genpose_amd.config import get_config
from genpose_amd.posenet_agent import PoseNet
from genpose_amd.weights_synth import make_state_dict
B, K, n = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 50, 100
sa = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=n)); sa.load_state_dict(make_state_dict(0, "score"))
pts = torch.from_numpy(synth.make_batch(B)).cuda(); cen = pts.mean(1)
sa.pred_func({"pts": pts, "pts_center": cen}, repeat_num=K, save_path=None); torch.cuda.synchronize()
smp = sa.net.last_sampler
feat = sa.net.pts_encoder(pts); cvec = sa.net.pose_score_net.cloud_embed(feat); x0 = torch.randn(B * K, 9, device="cuda")
def series(name, fn, reps=12):
    ts = []
    for i in range(reps):
        torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize(); ts.append(round((time.time() - t) * 1e3, 1))
    print(f"{name:28s}", ts)
series("encoder", lambda: sa.net.pts_encoder(pts))
series("graph replay", lambda: smp.graph.replay())
series("normal_ x2", lambda: (smp.z1.normal_(), smp.z2.normal_()))
series("normal_ x2 + replay", lambda: (smp.z1.normal_(), smp.z2.normal_(), smp.graph.replay()))
series("smp.run", lambda: smp.run(cvec, cen, x0))
series("encoder + smp.run", lambda: (sa.net.pts_encoder(pts), smp.run(cvec, cen, x0)))
series("pred_func", lambda: sa.pred_func({"pts": pts, "pts_center": cen}, repeat_num=K, save_path=None))
fixed = torch.randn(B * K, 9) * 50
saved = sa.net.prior_fn
sa.net.prior_fn = lambda shape, **k: fixed
series("pred_func, fixed prior", lambda: sa.pred_func({"pts": pts, "pts_center": cen}, repeat_num=K, save_path=None))
sa.net.prior_fn = saved
torch.set_num_threads(1)
series("pred_func, 1 CPU thread", lambda: sa.pred_func({"pts": pts, "pts_center": cen}, repeat_num=K, save_path=None))
