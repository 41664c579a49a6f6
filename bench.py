#!/usr/bin/env python
"""Benchmark of the GenPose inference hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched through torch.distributed.run)

One *step* = one pass of the hot path over one batch of synthetic clouds resident in HBM:
    PointNet++ encoder (B clouds x 1024 pts)  ->  per-cloud embedding  ->  K=50 candidates x 100-step
    predictor-corrector sampler (score network evaluated 100 times per candidate)  ->  pred_pose [B,50,9]
(BASELINE.json configs[1]: "1xMI355X: batch 64 clouds x 1024 pts, 50 candidates, 100 SDE steps, ScoreNet only";
PC-100 is the sampler whose NFE equals the step count).  `--pipeline full` adds the energy network,
ranking and top-60% aggregation (configs[2] shape).  Metric: poses/sec (one pose = one cloud's K-candidate estimate),
whole-job aggregate over all ranks; weak scaling (every rank owns its own B clouds; the only collective is the final
all-gather of the results over RCCL).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_SCORE_ROW = 0.5335e6  # minimal ("hoisted") FLOPs per pose row per score evaluation
FLOP_ENCODER = 2.201e9     # per cloud per encoder pass
FLOP_CLOUD_EMBED = 1.573e6
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clouds per GPU per step")
    ap.add_argument("--cand", type=int, default=50)
    ap.add_argument("--sde-steps", type=int, default=100)
    ap.add_argument("--sampler", choices=["pc", "ode"], default="pc")
    ap.add_argument("--pipeline", choices=["score", "full"], default="score")
    ap.add_argument("--no-pipeline", action="store_true", help="run the steps strictly one after another on one stream")
    ap.add_argument("--batches-per-launch", type=int, default=5,
                    help="pipelined PC workload: consecutive batches that share one encoder pass and one sampler launch chain "
                         "(the sampler's batch-global coupling stays per batch); 1 = one batch per launch")
    ap.add_argument("--overlap", action="store_true",
                    help="run the encoder of the next launch group on a second HIP stream under the sampler graph of the current one "
                         "(pays off with 1-2 batches per launch: 16.9 k vs 15.4 k poses/s at 1; no gain at 5, where both stages fill the chip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clouds", type=int, default=4)
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    # GP_BENCH_ONE_DEVICE=1 (self-test on a 1-GPU box): every rank uses cuda:0 and the collectives run on gloo
    one_dev = os.environ.get("GP_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("GP_BENCH_FORCE_DIST") == "1":  # GP_BENCH_FORCE_DIST: exercise RCCL with one rank (self-test)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" = RCCL on ROCm

    from genpose_amd import reward, synth
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.weights_synth import make_state_dict

    B, K, n = args.batch, args.cand, args.sde_steps
    steps = n if args.sampler == "pc" else None
    score_agent = PoseNet(get_config(device=str(dev), posenet_mode="score", sampler_mode=[args.sampler], sampling_steps=steps))
    score_agent.load_state_dict(make_state_dict(0, "score"))
    energy_agent = None
    if args.pipeline == "full":
        energy_agent = PoseNet(get_config(device=str(dev), posenet_mode="energy"))
        energy_agent.load_state_dict(make_state_dict(0, "energy"))

    # inputs resident in HBM before the timed region: this rank's B clouds (weak scaling: distinct clouds per rank)
    pts = torch.from_numpy(synth.make_batch(B, start=rank * B)).to(dev)
    centre = pts.mean(dim=1)
    T0 = 0.55

    def step():
        data = {"pts": pts, "pts_center": centre}
        pred = score_agent.pred_func(data, repeat_num=K, save_path=None, T0=T0)
        out = pred
        if energy_agent is not None:
            energy = energy_agent.get_energy(data={"pts": pts, "pts_center": centre}, pose_samples=pred, T=1e-5)
            out = reward.rank_aggregate(pred, energy, ratio=0.6)["avg_pose"]
        if dist is not None:  # the path's only exchange: gather every rank's result
            outs = [torch.empty_like(out) for _ in range(world)]
            dist.all_gather(outs, out.contiguous())
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Score-only PC workload: consecutive steps are software-pipelined over two HIP streams (encoder of step i+1 under the
    # sampler graph of step i, genpose_amd/pipeline.py); every step still runs completely inside the timed region.
    pipelined = args.sampler == "pc" and args.pipeline == "score" and not args.no_pipeline
    pipe = None
    if pipelined:
        from genpose_amd.pipeline import PipelinedPCPredictor
        pipe = PipelinedPCPredictor(score_agent, B, K, n, batches_per_launch=args.batches_per_launch, overlap=args.overlap)
    ode_grouped = args.sampler == "ode" and args.pipeline == "score" and not args.no_pipeline and args.batches_per_launch > 1
    ode_pred = None
    if ode_grouped:
        from genpose_amd.pipeline import GroupedODEPredictor
        ode_pred = GroupedODEPredictor(score_agent, B, K, T0=T0, batches_per_launch=args.batches_per_launch)
    G = args.batches_per_launch if (pipelined or ode_grouped) else 1

    def run_steps(count):
        if ode_grouped:
            outs = ode_pred.run([pts] * count)
            if dist is not None:
                for o in outs:
                    gathered = [torch.empty_like(o) for _ in range(world)]
                    dist.all_gather(gathered, o.contiguous())
            return
        if not pipelined:
            for _ in range(count):
                step()
            return
        outs = pipe.run([pts] * count)
        if dist is not None:  # the path's only exchange: gather every rank's result
            for o in outs:
                gathered = [torch.empty_like(o) for _ in range(world)]
                dist.all_gather(gathered, o)

    step()  # builds samplers / captures graphs outside the timed region
    if pipelined or ode_grouped:
        for g in sorted({G, args.warmup % G, args.steps % G} - {0}, reverse=True):
            run_steps(g)  # captures the G-batch graph and the graph of the ragged tail this run will meet
    run_steps(args.warmup)
    barrier()
    if pipe is not None:
        pipe.timing = True
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_steps(args.steps)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- roofline of the dominant kernel (pc_step: fused PC update + score network), HIP events on the launch stream
    roofline = None
    nfev = n
    if args.sampler == "pc":
        smp = pipe._sampler(0, G) if pipe is not None else score_agent.net._samplers[("pc", B, K, n, False)]
        reps = 5
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            smp.graph.replay()  # graph = exactly n+1 pc_step launches, nothing else
        e1.record()
        torch.cuda.synchronize()
        per_launch_s = e0.elapsed_time(e1) * 1e-3 / (reps * (n + 1))
        # `achieved` uses the kernel's own duration: HIP events around replays of the sampler graph on its launch stream
        # with nothing else in flight (this is also what a rocprofv3 kernel trace of this command reports, because the
        # profiler serialises the two streams: profiles/r1_bench_kernel_stats.csv).  In the pipelined timed region the
        # launches share the chip with the encoder of the next step, so their in-situ duration is longer; it is
        # reported next to it (events around every graph replay inside the timed region).
        flops_per_launch = G * B * K * FLOP_SCORE_ROW  # one launch serves G batches
        ach = flops_per_launch / per_launch_s / 1e12
        roofline = {"bound": "mfma", "kernel": f"pc_step_kernel<{smp.tile}>", "rows_per_launch": G * B * K, "achieved": round(ach, 2),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                    "avg_launch_us": round(per_launch_s * 1e6, 2), "flops_per_launch": flops_per_launch}
        # HBM-side bytes per launch come from the PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE in separate
        # rocprofv3 runs, gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md); only valid for the profiled shape
        tpath = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))[f"{roofline['kernel']}@{G * B * K}"]
                roofline["traffic"] = tj["corrected_bytes_per_launch"]
                roofline["traffic_note"] = (f"PMC, profiles/r1_pmc_traffic.json: raw {tj['raw_bytes_per_launch']} B, algorithmic "
                                            f"{tj['algorithmic_bytes_per_launch']} B; Infinity-Cache hits are counted (the 1 MB weight set "
                                            "is re-fetched by each of the 8 XCD L2s every launch)")
            except (KeyError, ValueError):
                pass
        in_situ = pipe.sampler_launch_seconds() if pipe is not None else None
        if in_situ:
            roofline["in_situ_avg_launch_us"] = round(in_situ * 1e6, 2)
            roofline["in_situ_achieved"] = round(flops_per_launch / in_situ / 1e12, 2)
    else:
        if ode_grouped:
            nfev = int(round(sum(ode_pred.last_nfev) / max(1, len(ode_pred.last_nfev))))
        else:
            st = score_agent.net._samplers[("ode", B, K)].last_stats
            nfev = int(st["nfev"])

    # the same workload with ONE batch per launch (no request batching), reported next to the headline for comparison
    one_batch = None
    if pipelined and G > 1 and world == 1:
        from genpose_amd.pipeline import PipelinedPCPredictor
        p1 = PipelinedPCPredictor(score_agent, B, K, n, batches_per_launch=1, overlap=False)
        p1.run([pts] * 3)
        torch.cuda.synchronize()
        nb = 10
        t1 = time.perf_counter()
        p1.run([pts] * nb)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        smp1 = p1._sampler(0, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            smp1.graph.replay()
        e1.record()
        torch.cuda.synchronize()
        l1 = e0.elapsed_time(e1) * 1e-3 / (5 * (n + 1))
        one_batch = {"value": round(B * nb / dt, 2), "unit": "poses/s", "ms_per_step": round(dt / nb * 1e3, 3), "kernel": f"pc_step_kernel<{smp1.tile}>",
                     "rows_per_launch": B * K, "avg_launch_us": round(l1 * 1e6, 2),
                     "frac": round(B * K * FLOP_SCORE_ROW / l1 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        del p1, smp1

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(args, K, n)

    if rank == 0:
        flop_per_pose = FLOP_ENCODER + FLOP_CLOUD_EMBED + K * nfev * FLOP_SCORE_ROW
        if energy_agent is not None:
            flop_per_pose += FLOP_ENCODER + FLOP_CLOUD_EMBED + K * FLOP_SCORE_ROW
        line = {
            "metric": "poses/sec (1024-pt cloud, 50 cand x 100 SDE steps)", "value": round(value, 2), "unit": "poses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[{2 if energy_agent is not None else 1}]: {B} clouds/GPU x 1024 pts, {K} candidates, "
                                   + (f"PC sampler {n} steps (NFE={n})" if args.sampler == "pc" else f"ODE sampler RK45 T0={T0} (NFE={nfev})")
                                   + (", ScoreNet only" if energy_agent is None else ", + EnergyNet ranking + top-60% aggregation"),
                       "clouds_per_gpu": B, "candidates": K, "sde_steps": n, "sampler": args.sampler, "pipeline": args.pipeline, "stream_pipelining": bool(pipelined and args.overlap), "batches_per_launch": G,
                       "weights": "seeded random (reference state-dict schema)", "parallelism": f"clouds sharded x{world}"},
            "whole_path_tflops": round(value * flop_per_pose / 1e12, 2),
            "gpu_event_ms_per_step": round(ev0.elapsed_time(ev1) / args.steps, 3),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "one_batch_per_launch": one_batch,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_cpu_baseline(args, K, n):
    """The oracle (CPU restatement of the reference path, validated against the imported reference) timed on the host
    cores of this box on a bounded sample of the same workload.  Checker infrastructure used as a baseline: allowed
    use of oracle/ (task statement §3)."""
    from genpose_amd import synth
    from oracle import genpose_oracle as go
    Bc = args.cpu_clouds
    sd = go.make_state_dict(0, "score")
    pts = torch.from_numpy(synth.make_batch(Bc, start=0))
    gen = torch.Generator().manual_seed(0)
    prior = torch.randn(Bc * K, 9, generator=gen)

    def once():
        if args.sampler == "pc":
            z1 = torch.randn(n, Bc * K, 9, generator=gen)
            z2 = torch.randn(n, Bc * K, 9, generator=gen)
            go.pred_func(sd, pts, pts.mean(dim=1), K, "pc", prior, sampling_steps=n, z_langevin=z1, z_predictor=z2)
        else:
            go.pred_func(sd, pts, pts.mean(dim=1), K, "ode", prior, T0=0.55)

    once()  # warm-up (library init, oneDNN primitives)
    t0 = time.perf_counter()
    reps = 0
    while True:
        once()
        reps += 1
        if time.perf_counter() - t0 > 10.0 or reps >= 20:
            break
    dt = time.perf_counter() - t0
    return {"value": round(Bc * reps / dt, 3), "unit": "poses/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} x ({Bc} clouds x 1024 pts, {K} cand, {args.sampler.upper()} {n} steps) = {dt:.1f} s of CPU work; "
                      "oracle/genpose_oracle.py (torch-CPU fp32 MLPs + OpenMP C ops), encoder + sampler end to end"}


if __name__ == "__main__":
    main()
