"""Host-side time of the sections of TrackingRunner.step (one sequence, 5 objects, steady state), without a profiler: where does the host spend the
frame, and is the GPU waiting for it?  Sections are timed with perf_counter (no synchronisation added); the GPU-side idle gap between graph A's
last kernel and the solver's first is measured with events."""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genpose_amd import runner, synth, reward, rotation  # noqa: E402
from genpose_amd.config import get_config  # noqa: E402
from genpose_amd.posenet_agent import PoseNet  # noqa: E402
from genpose_amd.weights_synth import make_state_dict  # noqa: E402
from genpose_amd.runner import add_noise_to_RT, _one_cpu_thread  # noqa: E402

sa = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"])); sa.load_state_dict(make_state_dict(0, "score"))
ea = PoseNet(get_config(posenet_mode="energy")); ea.load_state_dict(make_state_dict(0, "energy"))
n_obj, K, nfr = 5, 50, 30
base = torch.from_numpy(synth.make_batch(n_obj, start=0))
gt = torch.eye(4).repeat(n_obj, 1, 1); gt[:, :3, 3] = base.mean(dim=1)
frames = [(base + 0.002 * f).cuda() for f in range(nfr)]
names = [f"o{j}" for j in range(n_obj)]
tr = runner.TrackingRunner(sa, ea, repeat_num=K, T0=0.15)
for f in range(10):
    tr.step(frames[f % nfr], names, gt)
torch.cuda.synchronize()
net = sa.net
G = tr._graphs
acc = {}
def mark(name, t):
    acc.setdefault(name, []).append((time.perf_counter() - t) * 1e6)
    return time.perf_counter()
gaps = []
for f in range(200):
    pts = frames[(10 + f) % nfr]
    t = time.perf_counter(); t_frame = t
    centre, cvec_s, cvec_e = G.embed(pts); t = mark("embed(): copy + replay A + replay A' on the side stream", t)
    eA = torch.cuda.Event(enable_timing=True); eA.record()
    with _one_cpu_thread():
        noised = add_noise_to_RT(gt.float().cpu()); t = mark("add_noise_to_RT on the CPU (drawn every frame, unused in steady state)", t)
    init_sRT = tr.buffer["pred_sRT"].float()
    init_x = torch.cat([init_sRT[:, :3, 0], init_sRT[:, :3, 1], init_sRT[:, :3, 3] - centre], dim=1); t = mark("init_x (slices, sub, cat)", t)
    prior = net._prior_to_device((n_obj * K, 9), T=tr.T0); t = mark("prior draw on the CPU + pinned staging + H2D", t)
    x0 = (prior.view(n_obj, K, 9) + init_x.float().unsqueeze(1)).view(n_obj * K, 9); t = mark("x0", t)
    smp = net._samplers.get(("ode", n_obj, K, None))
    eS = torch.cuda.Event(enable_timing=True); eS.record()
    _, x = smp.run(cvec_s, centre, x0, tr.T0, num_steps=net.cfg.sampling_steps, eps=net.sampling_eps); t = mark("ODESampler.run (incl. the status read = the frame's one sync)", t)
    pred = x.reshape(n_obj, K, 9)
    energy, sorted_RTs, average_sRT = G.rank(pred, centre, cvec_e); t = mark("rank(): 3 copies + replay B", t)
    energy, sorted_RTs, average_sRT = energy.clone(), sorted_RTs.clone(), average_sRT.clone(); t = mark("3 clones", t)
    tr.buffer = {"model_name": list(names), "pred_sRT": average_sRT}
    acc.setdefault("frame (host)", []).append((time.perf_counter() - t_frame) * 1e6)
    torch.cuda.synchronize()
    gaps.append(eA.elapsed_time(eS) * 1e3)
for k, v in acc.items():
    print(f"{k:80s} median {statistics.median(v):8.1f} us   p90 {sorted(v)[int(0.9 * len(v))]:8.1f}")
print(f"GPU time between the event recorded right behind graph A's launch and the one recorded right before the solver's launches: median {statistics.median(gaps):.1f} us "
      "(graph A's kernels take ~520 us: anything above is the GPU waiting for the host)")
