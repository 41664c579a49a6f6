"""Latency-regime launch plans (round 5): 16-row tiles vs the head-split plan (three workgroups per tile, one head each).
PC step launch time and RK45 attempt time per row count; one tracking sequence frame by frame; BASELINE configs[0].
    python scratch/headsplit_plans.py > profiles/r5_plans.txt"""
import sys, time
sys.path.insert(0, ".")
import torch
from genpose_amd import synth
from genpose_amd.config import get_config
from genpose_amd.posenet_agent import PoseNet
from genpose_amd.runner import TrackingRunner
from genpose_amd.samplers import ODESampler, PCSampler
from genpose_amd.scorenet import ScoreNetHIP
from genpose_amd.weights_synth import make_state_dict

HS = 0x100
sd = make_state_dict(0, "score")
net = ScoreNetHIP(sd, "cuda")
FLOP = 0.5335e6


def ev_time(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    a.record(); [fn() for _ in range(reps)]; b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


print("# PC step (one launch of the captured chain = previous step's update + score evaluation), K = 50, 40-step chains, us per launch")
print(f"{'clouds':>6} {'rows':>6} {'tiles':>5} | {'16-row tiles':>12} {'head-split':>11} | auto")
for B in (1, 2, 5, 10, 16, 20, 27, 28, 32, 48, 64):
    K, n = 50, 40
    row = []
    for plan in (16, 16 | HS):
        smp = PCSampler(net, B, K, n, "cuda", tile=plan)
        cvec, cen, x0 = torch.randn(B, 768, device="cuda"), torch.randn(B, 3, device="cuda"), torch.randn(B * K, 9, device="cuda") * 50
        smp.run(cvec, cen, x0)
        us = ev_time(lambda: smp.graph.replay(), 20) / (n + 1)
        row.append(us)
    auto = PCSampler(net, B, K, n, "cuda").plan
    print(f"{B:6d} {B * K:6d} {(B * K + 15) // 16:5d} | {row[0]:9.1f} us {row[1]:8.1f} us | {'head-split' if auto & HS else auto}")

print("\n# RK45 attempt (6 stage evaluations + controller + stage-time embedding), T0 = 0.15 warm start, us per attempt (graph replay of the solve / attempts launched)")
print(f"{'clouds':>6} {'rows':>6} | {'16-row tiles (fused attempt kernel)':>36} {'head-split (6 stage launches)':>30}")
for B in (1, 3, 5, 6, 10, 20, 27):
    K = 50 if B > 1 else 10
    row = []
    for plan in (16, 16 | HS):
        smp = ODESampler(net, B, K, "cuda", tile=plan)
        cvec, cen = torch.randn(B, 768, device="cuda"), torch.randn(B, 3, device="cuda")
        x0 = torch.randn(B * K, 9, device="cuda") * 0.04
        for _ in range(3):
            smp.run(cvec, cen, x0, 0.15)
        t = time.perf_counter()
        reps = 20
        for _ in range(reps):
            smp.run(cvec, cen, x0, 0.15)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
        row.append((ms, smp.last_replays["attempts_launched"], int(smp.last_stats["nfev"])))
    print(f"{B:6d} {B * K:6d} | {row[0][0]:8.3f} ms per solve ({row[0][1]} attempts, nfev {row[0][2]}) {row[1][0]:8.3f} ms per solve ({row[1][1]} attempts, nfev {row[1][2]})")

print("\n# one tracking sequence, frame by frame (5 objects, K = 50, T0 = 0.15), 100 frames")
sa = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"])); sa.load_state_dict(sd)
ea = PoseNet(get_config(posenet_mode="energy")); ea.load_state_dict(make_state_dict(0, "energy"))
n_obj, K, nfr, warm = 5, 50, 106, 6
base = torch.from_numpy(synth.make_batch(n_obj, start=0))
gt = torch.eye(4).repeat(n_obj, 1, 1); gt[:, :3, 3] = base.mean(dim=1)
frames = [(base + 0.002 * (f % 30)).cuda() for f in range(nfr)]
names = [f"o{j}" for j in range(n_obj)]
import genpose_amd.samplers as S
for label, force in (("head-split (auto)", 0), ("16-row tiles (forced)", 16)):
    orig = S.ODESampler.__init__
    if force:
        def init(self, *a, _o=orig, **k):
            k.setdefault("tile", 16)
            _o(self, *a, **k)
        S.ODESampler.__init__ = init
    sa.net._samplers.clear()
    tr = TrackingRunner(sa, ea, repeat_num=K, T0=0.15)
    for f in range(warm): tr.step(frames[f], names, gt)
    torch.cuda.synchronize(); t = time.time()
    for f in range(warm, nfr): tr.step(frames[f], names, gt)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"{label:24s}: {dt / (nfr - warm) * 1e3:.3f} ms per frame; plan {sa.net.last_sampler.plan}; replays {sa.net.last_sampler.last_replays}")
    S.ODESampler.__init__ = orig

print("\n# BASELINE configs[0]: one cloud, 10 candidates, agent API (pred_func -> get_energy -> rank_aggregate), ms per call")
from genpose_amd import reward
pts = torch.from_numpy(synth.make_batch(1, start=4242)).cuda()
for sampler, steps, T0 in (("pc", 20, None), ("ode", None, 0.55)):
    for label, force in (("head-split (auto)", 0), ("16-row tiles (forced)", 16)):
        a = PoseNet(get_config(posenet_mode="score", sampler_mode=[sampler], sampling_steps=steps)); a.load_state_dict(sd)
        origs = (S.ODESampler.__init__, S.PCSampler.__init__)
        if force:
            def oinit(self, *aa, _o=origs[0], **k):
                k.setdefault("tile", 16); _o(self, *aa, **k)
            def pinit(self, *aa, _o=origs[1], **k):
                k.setdefault("tile", 16); _o(self, *aa, **k)
            S.ODESampler.__init__, S.PCSampler.__init__ = oinit, pinit
        def call():
            data = {"pts": pts, "pts_center": pts.mean(dim=1)}
            pred = a.pred_func(data, repeat_num=10, save_path=None, T0=T0)
            e = ea.get_energy(data=data, pose_samples=pred, T=1e-5)
            return reward.rank_aggregate(pred, e, ratio=0.6)
        for _ in range(10): call()  # (first call launch by launch, second captures the encoder passes, samplers capture on their first run)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(30): call()
        torch.cuda.synchronize()
        print(f"{sampler:3s} {label:24s}: {(time.perf_counter() - t) / 30 * 1e3:.3f} ms per call (plan {a.net.last_sampler.plan})")
        S.ODESampler.__init__, S.PCSampler.__init__ = origs
