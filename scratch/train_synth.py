"""Leaves the random-weight regime: trains the score model and the energy model on synthetic posed clouds (synth.make_posed_cloud) with
the training step that is pinned to the reference (genpose_amd/training.py, G14 / G15) - NOT a port of runners/trainer.py: a bounded
loop of Trainer.train_func, a learning-rate decay every --decay-every steps standing in for the reference's per-epoch decay.

    python scratch/train_synth.py --minutes-score 8 --minutes-energy 5 --out gpurun_out/trained

Writes <out>/ckpt_score.pth and <out>/ckpt_energy.pth (reference layout: {'clock', 'model_state_dict'} with the averaged weights, what
PoseNet.load_ckpt reads) and prints ms per step.  The split of a step between the hand-written grouping kernels and stock autograd comes
from rocprofv3 --kernel-trace --stats over a short run of this script (--steps), see scratch/r6_training_profile.sh.
"""
import argparse
import math
import os
import sys
import time
from multiprocessing import Pool

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genpose_amd import synth  # noqa: E402


DEVICE = "cuda"  # the grouping operators of the training step exist on the device only


def _chunk(idx):
    return synth.posed_batch(idx)


def dataset(n, start=0, workers=None):
    """n posed clouds generated side by side on the host cores -> dict of device tensors."""
    workers = workers or max(1, min(64, (os.cpu_count() or 8) - 2))
    chunks = [range(start + s, start + min(n, s + 256)) for s in range(0, n, 256)]
    with Pool(workers) as pool:
        parts = pool.map(_chunk, chunks)
    cat = {k: np.concatenate([p[k] for p in parts], 0) for k in parts[0]}
    return {k: torch.from_numpy(v).to(DEVICE) for k, v in cat.items() if k in ("pts", "gt_pose", "cat", "handle_visibility")}


def batch(ds, bs, gen):
    """A random batch with a random camera roll (rotation of the whole scene about the optical axis: clouds, R and t turn together)."""
    n = ds["pts"].shape[0]
    i = torch.randint(0, n, (bs,), device=DEVICE, generator=gen)
    a = (torch.rand(bs, device=DEVICE, generator=gen) * 2 - 1) * math.pi
    c, s, z, o = torch.cos(a), torch.sin(a), torch.zeros_like(a), torch.ones_like(a)
    Rz = torch.stack([c, -s, z, s, c, z, z, z, o], dim=-1).reshape(bs, 3, 3)
    pts = ds["pts"][i] @ Rz.transpose(1, 2)
    gt = (ds["gt_pose"][i].reshape(bs, 3, 3) @ Rz.transpose(1, 2)).reshape(bs, 9)
    centre = pts.mean(dim=1)
    zm = gt.clone()
    zm[:, 6:] -= centre
    return {"pts": pts, "zero_mean_pts": pts - centre.unsqueeze(1), "pts_center": centre, "gt_pose": gt, "zero_mean_gt_pose": zm,
            "id": ds["cat"][i], "handle_visibility": ds["handle_visibility"][i]}


def loop(tr, ds, gen, bs, gf_mode, seconds, max_steps, decay_every, candidates=None, tag=""):
    t0, times, losses = time.time(), [], []
    while time.time() - t0 < seconds and len(times) < max_steps:
        if tr.clock["step"] < tr.warmup or (tr.clock["step"] % decay_every == 0 and tr.clock["step"] > 0):
            tr.update_learning_rate()
        data = batch(ds, bs, gen)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = tr.train_func(data, pose_samples=None if candidates is None else candidates(data), gf_mode=gf_mode)
        e1.record()
        tr.tick()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
        losses.append({k: float(v.detach()) for k, v in out.items()})
        if len(times) % 100 == 0:
            last = {k: round(float(np.mean([l[k] for l in losses[-100:]])), 4) for k in losses[-1]}
            print(f"[{tag}] step {len(times)} ({time.time() - t0:.0f} s) lr {tr.optimizer.param_groups[-1]['lr']:.2e} loss {last}", flush=True)
    if times:
        skip = min(5, len(times) - 1)
        print(f"[{tag}] {len(times)} steps of gf_mode='{gf_mode}', batch {bs} x repeat {tr.repeat_num}: median {np.median(times[skip:]):.1f} ms/step, "
              f"mean {np.mean(times[skip:]):.1f}, wall {1e3 * (time.time() - t0) / len(times):.1f} ms/step incl. batch assembly", flush=True)
    return losses


def save(tr, path):
    torch.save({"clock": dict(tr.clock), "model_state_dict": tr.state_dict(ema=True)}, path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes-score", type=float, default=8.0)
    ap.add_argument("--minutes-energy", type=float, default=5.0)
    ap.add_argument("--ranking-fraction", type=float, default=0.3, help="share of the energy budget spent with the ranking loss (gf_mode 'energy')")
    ap.add_argument("--steps", type=int, default=10 ** 9, help="cap per phase (profiling runs)")
    ap.add_argument("--clouds", type=int, default=49152)
    ap.add_argument("--batch", type=int, default=192)       # scripts/train_energy.sh, configs/config.py
    ap.add_argument("--repeat", type=int, default=20)       # --repeat_num of the reference's training (configs/config.py)
    ap.add_argument("--decay-every", type=int, default=50)
    ap.add_argument("--out", default="gpurun_out/trained")
    ap.add_argument("--resume-score", default=None)
    a = ap.parse_args()
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.training import Trainer
    os.makedirs(a.out, exist_ok=True)
    torch.manual_seed(0)
    gen = torch.Generator(device=DEVICE).manual_seed(0)
    t = time.time()
    ds = dataset(a.clouds)
    print(f"{a.clouds} posed clouds in {time.time() - t:.1f} s on {os.cpu_count()} host cores", flush=True)

    score = Trainer(device=DEVICE, posenet_mode="score", repeat_num=a.repeat)
    if a.resume_score:
        score.load_ckpt(a.resume_score, load_model_only=True)
    loop(score, ds, gen, a.batch, "score", 60 * a.minutes_score, a.steps, a.decay_every, tag="score")
    save(score, os.path.join(a.out, "ckpt_score.pth"))
    score.save_ckpt("/tmp/ckpt_score_full.pth")  # the reference's full dictionary (+ optimiser, scheduler): exercised, not shipped (3 x the size)

    energy = Trainer(device=DEVICE, posenet_mode="energy", repeat_num=a.repeat)
    n_rank = a.ranking_fraction * a.minutes_energy
    loop(energy, ds, gen, a.batch, "energy_wo_ranking", 60 * (a.minutes_energy - n_rank), a.steps, a.decay_every, tag="energy w/o ranking")
    # candidates of the TRAINED score model from the HIP agent, five per cloud, as runners/trainer.py:355 draws them
    agent = PoseNet(get_config(posenet_mode="score", sampler_mode=["ode"], sampling_steps=None))
    agent.load_ckpt(model_dir=os.path.join(a.out, "ckpt_score.pth"), model_path=True, load_model_only=True)
    cand = lambda data: agent.pred_func(data={k: v for k, v in data.items()}, repeat_num=5, save_path=None).float()
    loop(energy, ds, gen, a.batch, "energy", 60 * n_rank, a.steps, a.decay_every, candidates=cand, tag="energy + ranking")
    save(energy, os.path.join(a.out, "ckpt_energy.pth"))


if __name__ == "__main__":
    main()
