import sys, time; sys.path.insert(0, '.')
import torch
from genpose_amd import synth
from genpose_amd.config import get_config
from genpose_amd.posenet_agent import PoseNet
from genpose_amd.weights_synth import make_state_dict
B, K, n = 256, 50, 100
sa = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=n)); sa.load_state_dict(make_state_dict(0, "score"))
pts = torch.from_numpy(synth.make_batch(B)).cuda(); cen = pts.mean(1)
sa.pred_func({"pts": pts, "pts_center": cen}, repeat_num=K, save_path=None); torch.cuda.synchronize()
ts = []
for i in range(12):
    t = time.time(); sa.pred_func({"pts": pts, "pts_center": cen}, repeat_num=K, save_path=None); t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    ts.append((round((t1 - t) * 1e3, 2), round((t2 - t) * 1e3, 2)))
print("pred_func (host return ms, synced ms):", ts)
rows = {"pts_feat": sa.net.pts_encoder(pts), "pts_center": cen, "_repeat": K}
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(6): sa.net.sample(rows, "pc", return_process=False)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
