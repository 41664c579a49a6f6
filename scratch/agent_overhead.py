"""Wall time of the agent-level calls (Level B drop-in) at B clouds vs the sum of their kernels: where the host costs sit."""
import sys, time; sys.path.insert(0, '.')
import torch
from genpose_amd import reward, synth
from genpose_amd.config import get_config
from genpose_amd.posenet_agent import PoseNet
from genpose_amd.weights_synth import make_state_dict
B, K, n = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 50, 100
sa = PoseNet(get_config(posenet_mode="score", sampler_mode=["pc"], sampling_steps=n)); sa.load_state_dict(make_state_dict(0, "score"))
ea = PoseNet(get_config(posenet_mode="energy")); ea.load_state_dict(make_state_dict(0, "energy"))
pts = torch.from_numpy(synth.make_batch(B)).cuda(); cen = pts.mean(1)
def T(fn, reps=5):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.time() - t) / reps * 1e3, r
t_pred, pred = T(lambda: sa.pred_func({"pts": pts, "pts_center": cen}, repeat_num=K, save_path=None))
t_en, en = T(lambda: ea.get_energy(data={"pts": pts, "pts_center": cen}, pose_samples=pred, T=1e-5))
t_rank, _ = T(lambda: reward.rank_aggregate(pred, en, ratio=0.6))
t_enc, _ = T(lambda: sa.net.pts_encoder(pts))
smp = sa.net._samplers[("pc", B, K, n, False)]
t_graph, _ = T(lambda: smp.graph.replay())
t_prior, _ = T(lambda: sa.net.prior_fn((B * K, 9)).to("cuda"))
print(f"B={B}: pred_func {t_pred:.2f} ms (encoder {t_enc:.2f} + sampler graph {t_graph:.2f} + prior {t_prior:.2f}), get_energy {t_en:.2f} ms, rank_aggregate {t_rank:.2f} ms")
rows = {"pts_feat": sa.net.pts_encoder(pts), "pts_center": cen, "_repeat": K}
t_sample, _ = T(lambda: sa.net.sample(rows, "pc", return_process=False))
cvec = sa.net.pose_score_net.cloud_embed(rows["pts_feat"]); x0 = torch.randn(B * K, 9, device="cuda")
t_run, _ = T(lambda: smp.run(cvec, cen, x0))
t_norm, _ = T(lambda: (smp.z1.normal_(), smp.z2.normal_()))
t_feat, _ = T(lambda: sa.net(({"pts": pts, "pts_center": cen}), mode="pts_feature"))
print(f"   net.sample {t_sample:.2f} ms, PCSampler.run {t_run:.2f} ms, normal_ x2 {t_norm:.2f} ms, pts_feature {t_feat:.2f} ms")
