#!/usr/bin/env python
"""Benchmark of the GenPose inference hot path on MI355X (contract: see the task statement / DESIGN.md §6).

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU, RCCL),
or - when no launcher environment is present - bench.py spawns exactly that command itself and relays rank 0's JSON line.

One *step* = one pass of the hot path over one batch of synthetic clouds resident in HBM:
    PointNet++ encoder (B clouds x 1024 pts)  ->  per-cloud embedding  ->  K=50 candidates x 100-step
    predictor-corrector sampler (score network evaluated 100 times per candidate)  ->  pred_pose [B,50,9]
(BASELINE.json configs[1]: "1xMI355X: batch 64 clouds x 1024 pts, 50 candidates, 100 SDE steps, ScoreNet only";
PC-100 is the sampler whose NFE equals the step count).  `--pipeline full` adds the energy network,
ranking and top-60% aggregation (configs[2] shape).  Metric: poses/sec (one pose = one cloud's K-candidate estimate),
whole-job aggregate over all ranks; weak scaling (every rank owns its own B clouds; the only collective is the final
all-gather of the results over RCCL).

Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize() on both sides and
reduced with MAX over ranks.  The K-step block is repeated until >= 1 s has been timed (`--repeats` fixes the count); `value` and
`ms_per_step` come from the MEDIAN block, the spread is reported next to them.
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_SCORE_ROW = 0.5335e6  # minimal ("hoisted") FLOPs per pose row per score evaluation
FLOP_ENCODER = 2.201e9     # per cloud per encoder pass, the reference's count (every first SA layer over the grouped [C+3] rows)


def encoder_flops_executed():
    """FLOPs the encoder EXECUTES per cloud after hoisting the feature half of every first SA layer: once per source
    point instead of once per (centre, sample) row.  Light config (pointnet2.py:57-66)."""
    levels = [  # (source points n, input channels, [(rows per cloud, c1, c2, c3) per scale])
        (1024, 0, [(512 * 16, 16, 16, 32), (512 * 32, 32, 32, 64)]),
        (512, 96, [(256 * 16, 64, 64, 128), (256 * 32, 64, 96, 128)]),
        (256, 256, [(128 * 16, 128, 196, 256), (128 * 32, 128, 196, 256)]),
        (128, 512, [(128, 256, 256, 512), (128, 256, 384, 512)]),
    ]
    tot = 0
    for n, cin, scales in levels:
        tot += 2 * n * cin * sum(c1 for _, c1, _, _ in scales)                      # hoisted feature half (point_linear / producer epilogue)
        tot += sum(rows * 2 * (3 * c1 + c1 * c2 + c2 * c3) for rows, c1, c2, c3 in scales)  # xyz half + layers 2-3 per grouped row
    return float(tot)


FLOP_ENCODER_EXECUTED = encoder_flops_executed()  # 1.69e9
FLOP_CLOUD_EMBED = 1.573e6
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
# Request batching of the default run: consecutive 64-cloud batches that share one encoder pass and one sampler launch chain (each keeps
# its own batch-global coupling and gets its stand-alone result).  Measured on MI355X: 1 -> 17.0 k, 5 -> 24.9 k, 10 -> 26.4 k, 20 -> 27.0 k
# poses/s (32 000 rows = 250 128-row workgroups of the chain-form sampler, one per CU; the encoder's persistent kernels amortise over 640 clouds).
DEFAULT_BATCHES_PER_LAUNCH = 10


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=0, help="timed K-step blocks (0 = as many as it takes to time >= 1 s, at most 64)")
    ap.add_argument("--batch", type=int, default=64, help="clouds per GPU per step")
    ap.add_argument("--cand", type=int, default=50)
    ap.add_argument("--sde-steps", type=int, default=100)
    ap.add_argument("--sampler", choices=["pc", "ode"], default="pc")
    ap.add_argument("--pipeline", choices=["score", "full"], default="score")
    ap.add_argument("--no-pipeline", action="store_true", help="run the steps strictly one after another on one stream")
    ap.add_argument("--batches-per-launch", type=int, default=DEFAULT_BATCHES_PER_LAUNCH,
                    help="pipelined PC workload: consecutive batches that share one encoder pass and one sampler launch chain "
                         "(the sampler's batch-global coupling stays per batch); 1 = one batch per launch")
    ap.add_argument("--overlap", action="store_true",
                    help="run the encoder of the next launch group on a second HIP stream under the sampler graph of the current one "
                         "(pays off with 1-2 batches per launch; no gain at 5, where both stages fill the chip)")
    ap.add_argument("--no-fps-ahead", action="store_true", help="do not run furthest point sampling of the next launch group on a side stream")
    ap.add_argument("--sampler-streams", type=int, default=1,
                    help="launch chains in flight (each on its own HIP stream; implies --overlap): with --batches-per-launch 1 two chains interleave "
                         "two 64-cloud batches' steps on the chip, every batch still its own chain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clouds", type=int, default=64, help="clouds of the CPU-baseline sample (BASELINE.md §3: one 64-cloud batch)")
    ap.add_argument("--cpu-budget", type=float, default=30.0, help="seconds of CPU work the baseline leg may spend")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[2] / ODE-100 / drop-in side measurements")
    ap.add_argument("--only-drop-in", action="store_true",
                    help="run only the `drop_in_eval_single` leg (the reference's eval_single call sequence through the agent API) and print it - "
                         "for rocprofv3 kernel statistics of that path")
    ap.add_argument("--only-split-bf16", action="store_true", help="run only the opt-in split-bf16 encoder leg (and print it)")
    ap.add_argument("--tracking", action="store_true",
                    help="BASELINE configs[4]: tracking mode - every rank streams --sequences whole sequences (warm-started candidates, PF-ODE "
                         "sampler from T0 = 0.15, energy ranking, aggregation per frame); a step = one frame of every sequence")
    ap.add_argument("--sequences", type=int, default=64, help="tracking: concurrent sequences per GPU")
    ap.add_argument("--objects", type=int, default=5, help="tracking: objects per frame")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher environment: run the documented launch line ourselves (one rank per GPU)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    env["GP_BENCH_LAUNCH"] = "self"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    got_line = False
    for ln in p.stdout:  # relay; remember whether rank 0 printed its JSON line
        got_line = got_line or ln.startswith("{")
        sys.stdout.write(ln)
        sys.stdout.flush()
    rc = p.wait()
    if not got_line:
        print(json.dumps({"metric": "poses/sec (1024-pt cloud, 50 cand x 100 SDE steps)", "value": None, "unit": "poses/s", "n_gpus": args.gpus,
                          "steps": args.steps, "warmup": args.warmup, "error": f"torch.distributed.run exited with code {rc} and rank 0 printed no line",
                          "launch": {"mode": "self", "world_size_env": args.gpus}}), flush=True)
    return rc


class _StartGuard:
    """The first multi-rank start must never end without a parseable line: whatever goes wrong between the launcher and the first
    completed collective (a rank that dies, a rendezvous that never completes, RCCL hanging in its first all-reduce, the launcher
    terminating the survivors of a failed peer), rank 0 prints ONE JSON line with an "error" field, the backend and what torch sees of
    the box - and every rank leaves, so that the launcher returns instead of sitting in the driver's time limit."""

    def __init__(self, args, rank, world):
        self.args, self.rank, self.world, self.backend, self.done, self.timer = args, rank, world, None, False, None

    def line(self, msg):
        torch = sys.modules.get("torch")  # (not imported here: this may run on the watcher thread while the main thread is still importing it)
        try:
            ndev = torch.cuda.device_count() if torch is not None else None
        except Exception:  # noqa: BLE001
            ndev = None
        return json.dumps({"metric": "poses/sec (1024-pt cloud, 50 cand x 100 SDE steps)", "value": None, "unit": "poses/s", "n_gpus": self.args.gpus,
                           "steps": self.args.steps, "warmup": self.args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                           "vs_baseline": None, "dtype": "f32", "data": "synthetic", "error": msg,
                           "launch": {"mode": os.environ.get("GP_BENCH_LAUNCH", "torch.distributed.run"), "world_size_env": self.world,
                                      "backend": self.backend, "device_count": ndev, "rank": self.rank,
                                      "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"},
                           "roofline": None, "cpu_baseline": None})

    def fail(self, msg, code=3):
        if not self.done:
            self.done = True
            if self.rank == 0:
                print(self.line(msg), flush=True)
            else:
                print(f"[bench.py rank {self.rank}] {msg}", file=sys.stderr, flush=True)
        os._exit(code)  # no atexit handlers, no destructors of a half-built process group

    def arm(self, seconds, what):
        import threading
        self.disarm()
        self.timer = threading.Timer(seconds, lambda: self.fail(f"{what}: no progress for {seconds:.0f} s"))
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def watch_sigterm(self):
        """SIGTERM from the launcher (it terminates the survivors when a peer rank has failed) must produce the line even while the main
        thread sits inside a C++ rendezvous or collective, where a Python-level signal handler would not run: the C-level handler writes
        the signal number to a wake-up pipe at once and a watcher thread takes it from there."""
        import signal
        import threading
        r, w = os.pipe()
        os.set_blocking(w, False)
        signal.signal(signal.SIGTERM, lambda *_: None)  # (a Python-level handler must exist for the wake-up write to happen)
        signal.set_wakeup_fd(w, warn_on_full_buffer=False)

        def watch():
            while True:
                b = os.read(r, 1)
                if b and b[0] == signal.SIGTERM:
                    self.fail("terminated by the launcher before the result line (a peer rank failed?)", code=4)
        threading.Thread(target=watch, daemon=True).start()


def main():
    args = parse()
    # GP_BENCH_FORCE_LAUNCH=1 (self-test): take the spawn path for one rank too
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("GP_BENCH_FORCE_LAUNCH") == "1"):
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the host driver only supports dmabuf IPC: without this RCCL's first exchange between processes fails with `hipIpcGetMemHandle: invalid
    # argument` (already exported on the boxes this runs on; set here too so that a bare launcher environment cannot lose it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    guard = _StartGuard(args, rank, world)
    if world > 1:
        guard.watch_sigterm()  # from the first moment on: a peer may die while this rank is still importing torch
    import torch
    if world != args.gpus:
        guard.fail(f"bench.py --gpus {args.gpus} under a launcher with WORLD_SIZE={world}", code=2)
    if os.environ.get("GP_BENCH_KILL_RANK") == str(rank):  # rehearsal of a rank that dies before the rendezvous (tests/test_gpu_bench.py)
        time.sleep(float(os.environ.get("GP_BENCH_KILL_AFTER", "3")))
        os._exit(17)
    # GP_BENCH_ONE_DEVICE=1 (self-test on a 1-GPU box): every rank uses cuda:0 and the collectives run on gloo
    one_dev = os.environ.get("GP_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        guard.fail(f"LOCAL_RANK {local_rank} but torch sees {torch.cuda.device_count()} device(s)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = None
    if world > 1 or os.environ.get("GP_BENCH_FORCE_DIST") == "1" or os.environ.get("GP_BENCH_LAUNCH") == "self":  # GP_BENCH_FORCE_DIST: exercise RCCL with one rank (self-test)
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = guard.backend = "gloo" if one_dev else "nccl"  # "nccl" = RCCL on ROCm
        limit = float(os.environ.get("GP_BENCH_START_TIMEOUT", "600"))  # (the first `import torch` on a fresh box takes minutes and differs between ranks)
        if world == 1:
            guard.watch_sigterm()
        guard.arm(limit + 30, f"process group start ({backend}, {world} ranks)")
        try:
            to = datetime.timedelta(seconds=limit)
            if one_dev:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=to)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=to)
            # the first collective (RCCL builds its rings here): every rank contributes 1, every rank must see the world size
            one = torch.ones(1, device="cpu" if one_dev else dev)
            dist.all_reduce(one)
            if not one_dev:
                torch.cuda.synchronize()
            seen = int(round(float(one.item())))
            if seen != world or dist.get_world_size() != world:
                guard.fail(f"first all-reduce saw {seen} rank(s), process group reports {dist.get_world_size()}, expected {world}")
        except SystemExit:
            raise
        except BaseException as exc:  # noqa: BLE001
            guard.fail(f"process group start failed on {backend}: {type(exc).__name__}: {exc}")
        guard.disarm()

    from genpose_amd import reward, synth
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.weights_synth import make_state_dict

    if args.only_drop_in:
        print(json.dumps({"drop_in_eval_single": drop_in_leg(torch, str(dev), args.cand)}), flush=True)
        return
    if args.tracking:
        tracking_bench(torch, args, dist, dev, world, rank, one_dev, backend)
        guard.done = True
        if dist is not None:
            dist.destroy_process_group()
        return
    B, K, n = args.batch, args.cand, args.sde_steps
    steps = n if args.sampler == "pc" else None
    score_agent = PoseNet(get_config(device=str(dev), posenet_mode="score", sampler_mode=[args.sampler], sampling_steps=steps))
    score_agent.load_state_dict(make_state_dict(0, "score"))
    energy_agent = None
    if args.pipeline == "full":
        energy_agent = PoseNet(get_config(device=str(dev), posenet_mode="energy"))
        energy_agent.load_state_dict(make_state_dict(0, "energy"))

    # inputs resident in HBM before the timed region (weak scaling: distinct clouds per rank).  Every batch of a timed block is a
    # DIFFERENT set of clouds (ball-query early exits are data dependent): a pool of `steps` batches, walked in order.
    npool = max(1, args.steps)
    pool = [torch.from_numpy(synth.make_batch(B, start=(rank * npool + j) * B)).to(dev) for j in range(npool)]
    pts = pool[0]
    centre = pts.mean(dim=1)
    T0 = 0.55
    batches_of = lambda count: [pool[j % npool] for j in range(count)]

    full_pred = None
    if energy_agent is not None and args.sampler == "pc" and not args.no_pipeline:
        from genpose_amd.pipeline import FullPipelinePredictor
        full_pred = FullPipelinePredictor(score_agent, energy_agent, B, K, n, batches_per_launch=min(args.batches_per_launch, 5))

    def gather(out):
        if dist is not None:  # the path's only exchange: gather every rank's result
            o = out.contiguous().cpu() if one_dev else out.contiguous()  # gloo (one-device self-test) gathers host tensors
            outs = [torch.empty_like(o) for _ in range(world)]
            dist.all_gather(outs, o)

    def step(pts=pts):
        if full_pred is not None:
            out = full_pred.run(pts)["avg_pose"]
            gather(out)
            return out
        data = {"pts": pts, "pts_center": pts.mean(dim=1)}
        pred = score_agent.pred_func(data, repeat_num=K, save_path=None, T0=T0)
        out = pred
        if energy_agent is not None:
            energy = energy_agent.get_energy(data=data, pose_samples=pred, T=1e-5)  # same dict: the grouping ticket is taken over
            out = reward.rank_aggregate(pred, energy, ratio=0.6)["avg_pose"]
        gather(out)
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Score-only PC workload: `batches_per_launch` consecutive batches share one encoder pass and one sampler launch chain
    # (genpose_amd/pipeline.py); every step still runs completely inside the timed region.
    pipelined = args.sampler == "pc" and args.pipeline == "score" and not args.no_pipeline
    pipe = None
    if pipelined:
        from genpose_amd.pipeline import PipelinedPCPredictor
        pipe = PipelinedPCPredictor(score_agent, B, K, n, batches_per_launch=args.batches_per_launch, overlap=args.overlap or args.sampler_streams > 1,
                                    fps_ahead=not args.no_fps_ahead, sampler_streams=args.sampler_streams, depth=max(2, args.sampler_streams))
    ode_grouped = args.sampler == "ode" and args.pipeline == "score" and not args.no_pipeline and args.batches_per_launch > 1
    ode_pred = None
    if ode_grouped:
        from genpose_amd.pipeline import GroupedODEPredictor
        ode_pred = GroupedODEPredictor(score_agent, B, K, T0=T0, batches_per_launch=args.batches_per_launch)
    G = args.batches_per_launch if (pipelined or ode_grouped) else (full_pred.G if full_pred is not None else 1)

    def run_steps(count):
        if full_pred is not None:
            for o in full_pred.run_many(batches_of(count)):
                gather(o["avg_pose"])
            return
        if ode_grouped:
            for o in ode_pred.run(batches_of(count)):
                gather(o)
            return
        if not pipelined:
            for j in range(count):
                step(pool[j % npool])
            return
        for o in pipe.run(batches_of(count)):
            gather(o)

    step()  # builds samplers / captures graphs outside the timed region
    if pipelined or ode_grouped or full_pred is not None:
        for g in sorted({G, args.warmup % G, args.steps % G} - {0}, reverse=True):
            run_steps(g)  # captures the G-batch graph and the graph of the ragged tail this run will meet
    run_steps(args.warmup)
    barrier()
    if pipe is not None:
        pipe.timing = True

    def timed_block():
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        run_steps(args.steps)
        ev1.record()
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], device="cpu" if one_dev else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, ev0.elapsed_time(ev1)

    blocks = [timed_block()]
    reps = args.repeats if args.repeats > 0 else max(1, min(64, math.ceil(1.0 / blocks[0][0])))  # identical on every rank (MAX-reduced time)
    while len(blocks) < reps:
        blocks.append(timed_block())
    times = sorted(b[0] for b in blocks)
    elapsed = statistics.median(times)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    gpu_event_ms = statistics.median(b[1] for b in blocks) / args.steps

    # ---- roofline of the dominant kernel (pc_step: fused PC update + score network), HIP events on the launch stream
    roofline = None
    nfev = n
    if args.sampler == "pc" and full_pred is None:
        smp = pipe._sampler(0, G) if pipe is not None else score_agent.net.last_sampler
        roofline = pc_roofline(torch, smp, G * B * K, n)
        in_situ = pipe.sampler_launch_seconds() if pipe is not None else None
        if in_situ:
            roofline["in_situ_avg_launch_us"] = round(in_situ * 1e6, 2)
    elif args.sampler == "ode":
        if ode_grouped:
            nfev = int(round(sum(ode_pred.last_nfev) / max(1, len(ode_pred.last_nfev))))
        else:
            nfev = int(score_agent.net.last_sampler.last_stats["nfev"])

    if args.only_split_bf16:
        print(json.dumps({"headline_f32": round(value, 2), "encoder_split_bf16": split_bf16_leg(torch, score_agent, pool, B, K, n, G, str(dev), False),
                          "encoder_and_sampler_split_bf16": split_bf16_leg(torch, score_agent, pool, B, K, n, G, str(dev), True)}), flush=True)
        return
    side = {}
    if rank == 0 and world == 1:
        # the same workload with ONE batch per launch (no request batching), reported next to the headline for comparison
        if pipelined and G > 1:
            side["one_batch_per_launch"] = one_batch_leg(torch, score_agent, pool, B, K, n)
        if not args.no_secondary and args.pipeline == "score" and args.sampler == "pc" and not args.no_pipeline:
            side["ode_100"] = ode_leg(torch, B, K, G, T0, pool, str(dev))
            side["full_pipeline_256"] = full_pipeline_leg(torch, K, n, str(dev))
            side["drop_in_eval_single"] = drop_in_leg(torch, str(dev), K)
            side["config0_single_object"] = config0_leg(torch, str(dev))
            side["energy_model_pc_step"] = energy_model_leg(torch, str(dev), B, K, G)
            # (the two opt-in split-bf16 legs are NOT part of the default run since round 6: frozen, `--only-split-bf16` prints them)
        if not args.no_cpu_baseline:
            side["cpu_baseline"] = run_cpu_baseline(torch, args, K, n)

    if rank == 0:
        flop_per_pose = FLOP_ENCODER + FLOP_CLOUD_EMBED + K * nfev * FLOP_SCORE_ROW
        flop_exec = FLOP_ENCODER_EXECUTED + FLOP_CLOUD_EMBED + K * nfev * FLOP_SCORE_ROW
        if energy_agent is not None:
            flop_per_pose += FLOP_ENCODER + FLOP_CLOUD_EMBED + K * FLOP_SCORE_ROW
            flop_exec += FLOP_ENCODER_EXECUTED + FLOP_CLOUD_EMBED + K * FLOP_SCORE_ROW
        line = {
            "metric": "poses/sec (1024-pt cloud, 50 cand x 100 SDE steps)", "value": round(value, 2), "unit": "poses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the full pipeline on more than one GPU is BASELINE configs[3] (2048 clouds = 8 x 256, sharded, results gathered)
            "config": {"workload": f"configs[{(3 if world > 1 else 2) if energy_agent is not None else 1}]: "
                                   + (f"{world * B} clouds sharded over {world} GPUs = " if world > 1 else "") + f"{B} clouds/GPU x 1024 pts, {K} candidates, "
                                   + (f"PC sampler {n} steps (NFE={n})" if args.sampler == "pc" else f"ODE sampler RK45 T0={T0} (NFE={nfev})")
                                   + (", ScoreNet only" if energy_agent is None else ", + EnergyNet ranking + top-60% aggregation"),
                       "clouds_per_gpu": B, "candidates": K, "sde_steps": n, "sampler": args.sampler, "pipeline": args.pipeline,
                       "stream_pipelining": bool(pipelined and (args.overlap or args.sampler_streams > 1)), "batches_per_launch": G,
                       "sampler_streams": args.sampler_streams,
                       "weights": "seeded random (reference state-dict schema)", "parallelism": f"clouds sharded x{world}"},
            "timing": {"blocks": len(blocks), "steps_per_block": args.steps, "timed_s": round(sum(times), 3), "block_ms_median": round(elapsed * 1e3, 3),
                       "block_ms_min": round(times[0] * 1e3, 3), "block_ms_max": round(times[-1] * 1e3, 3), "statistic": "median block"},
            "launch": {"mode": os.environ.get("GP_BENCH_LAUNCH", "direct" if world == 1 else "torch.distributed.run"),
                       "world_size_observed": (dist.get_world_size() if dist is not None else 1), "backend": backend},
            # whole_path_tflops credits the encoder with the reference's 2.201 GFLOP/cloud (SURVEY §8d's canonical figure);
            # executed_tflops counts what the kernels actually execute (first SA layers hoisted: 1.69 GFLOP/cloud)
            "whole_path_tflops": round(value * flop_per_pose / 1e12, 2),
            "executed_tflops": round(value * flop_exec / 1e12, 2),
            "executed_frac_of_f32_mfma_peak": round(value * flop_exec / 1e12 / (world * PEAK_F32_MFMA_TFLOPS), 4),
            "gpu_event_ms_per_step": round(gpu_event_ms, 3),
            "roofline": roofline, "cpu_baseline": side.pop("cpu_baseline", None),
        }
        line.update(side)
        guard.done = True
        print(json.dumps(line), flush=True)
    guard.done = True
    if dist is not None:
        dist.destroy_process_group()


def tracking_bench(torch, args, dist, dev, world, rank, one_dev, backend):
    """BASELINE configs[4] (evaluation_tracking.py path): the path does not shard below a sequence (frames are sequential: warm start),
    so every rank owns WHOLE sequences (dist.ShardedTracking: replicas only, no data-path collective) and drives them in lock-step
    through one MultiSequenceTracker - the frames its sequences are at share every launch (one encoder pass, one device-resident RK45
    solve with a step controller per sequence, one energy pass, one ranking launch).  A step = one frame of every sequence of the
    rank; value = poses (objects x frames) per second over all ranks; the per-frame results are gathered once per timed block."""
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.dist import ShardedTracking
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.runner import MultiSequenceTracker
    from genpose_amd.weights_synth import make_state_dict
    S, n_obj, K, T0, nfr = args.sequences, args.objects, args.cand, 0.15, 30
    sa = PoseNet(get_config(device=str(dev), posenet_mode="score", sampler_mode=["ode"]))
    sa.load_state_dict(make_state_dict(0, "score"))
    ea = PoseNet(get_config(device=str(dev), posenet_mode="energy"))
    ea.load_state_dict(make_state_dict(0, "energy"))
    st = ShardedTracking(lambda nloc: MultiSequenceTracker(sa, ea, nloc, repeat_num=K, T0=T0, max_objects_per_frame=max(8, n_obj)), world * S)
    seqs = []
    for sq in st.owned():  # REAL275-shaped synthetic sequences: n_obj objects drifting 2 mm per frame, 30 frames, then the sequence restarts
        base = torch.from_numpy(synth.make_batch(n_obj, start=50 * sq))
        gt = torch.eye(4).repeat(n_obj, 1, 1)
        gt[:, :3, 3] = base.mean(dim=1)
        seqs.append(([(base + 0.002 * f).to(dev) for f in range(nfr)], [f"s{sq}o{j}" for j in range(n_obj)], gt))
    clock = [0]

    def frame_step():
        f = clock[0] % nfr
        if f == 0:
            st.tracker.reset()  # a new 30-frame sequence: no warm start across the boundary
        out = st.tracker.step([(q[0][f], q[1], q[2]) for q in seqs])
        clock[0] += 1
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def gather(out):
        if dist is not None:  # the path's only exchange: the per-frame results of every sequence
            o = torch.stack([r["average_sRT"] for r in out]).contiguous()
            o = o.cpu() if one_dev else o
            dist.all_gather([torch.empty_like(o) for _ in range(world)], o)

    for _ in range(max(2, args.warmup)):
        out = frame_step()
    barrier()

    def timed_block():
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = frame_step()
        gather(out)
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], device="cpu" if one_dev else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, [int(r["nfev"]) for r in out]

    blocks = [timed_block()]
    reps = args.repeats if args.repeats > 0 else max(1, min(16, math.ceil(1.0 / blocks[0][0])))
    while len(blocks) < reps:
        blocks.append(timed_block())
    times = sorted(b[0] for b in blocks)
    elapsed = statistics.median(times)
    nfev = statistics.median(blocks[-1][1])
    single = None
    if rank == 0 and world == 1 and not args.no_secondary:
        # latency of ONE sequence processed frame by frame (runner.TrackingRunner: a frame = two hipGraphs around the adaptive solve, the
        # energy model's encoder underneath the solve) - what evaluation_tracking.py's loop does, one call per frame
        from genpose_amd.runner import TrackingRunner
        tr = TrackingRunner(sa, ea, repeat_num=K, T0=T0)
        frames_1, names_1, gt_1 = seqs[0]
        for f in range(8):
            tr.step(frames_1[f % nfr], names_1, gt_1)
        torch.cuda.synchronize()
        nf, per, nfe = 60, [], []
        for rep in range(7):
            t0 = time.perf_counter()
            evals = 0
            for f in range(nf):
                tr.step(frames_1[(8 + rep * nf + f) % nfr], names_1, gt_1)
                evals += int(sa.net.last_sampler.last_stats["nfev"])  # (host-side: the solve's status read has already brought it over)
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / nf)
            nfe.append(evals / nf)
        dt1 = statistics.median(per)
        single = {"ms_per_frame": round(dt1 * 1e3, 3), "ms_per_frame_min": round(min(per) * 1e3, 3), "ms_per_frame_max": round(max(per) * 1e3, 3),
                  "repeats": len(per), "frames_per_repeat": nf, "statistic": "median of the repeats", "ms_per_frame_by_repeat": [round(v * 1e3, 3) for v in per],
                  # the adaptive solve takes 5-8 attempts per frame depending on where the (random-weight) tracker has wandered: the spread of
                  # the repeats is this workload spread, not timer noise (scratch/track_clock_probe.py: the clock stays at 2.39 GHz)
                  "mean_nfev_per_frame_by_repeat": [round(v, 1) for v in nfe],
                  "frames_per_s": round(1.0 / dt1, 1), "objects_per_frame": n_obj, "nfev": int(sa.net.last_sampler.last_stats["nfev"]),
                  "workload": "one sequence, one TrackingRunner.step per frame"}
        try:  # the same loop on the TRAINED checkpoints of tests/golden/trained (a tracker that tracks: the solve takes ~4 attempts per frame)
            single["trained_weights"] = single_sequence_trained(torch, str(dev), K, T0, n_obj)
        except Exception as exc:  # noqa: BLE001  (an informative extra must never cost the line)
            single["trained_weights"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        value = world * S * n_obj * args.steps / elapsed
        flop_per_pose = 2 * (FLOP_ENCODER + FLOP_CLOUD_EMBED) + K * (nfev + 1) * FLOP_SCORE_ROW
        print(json.dumps({
            "metric": "poses/sec (1024-pt cloud, 50 cand, tracking: warm-started PF-ODE)", "value": round(value, 2), "unit": "poses/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[4]: tracking, {S} sequences/GPU x 30 frames, {n_obj} objects/frame x 1024 pts, {K} candidates, "
                                   f"PF-ODE RK45 from T0={T0} (NFE={nfev}) + EnergyNet ranking + top-60% aggregation per frame",
                       "sequences_per_gpu": S, "objects_per_frame": n_obj, "candidates": K, "sampler": "ode", "T0": T0, "nfev": nfev,
                       "frames_per_s": round(world * S * args.steps / elapsed, 1), "weights": "seeded random (reference state-dict schema)",
                       "parallelism": f"whole sequences per rank x{world} (replicas only)"},
            "timing": {"blocks": len(blocks), "steps_per_block": args.steps, "timed_s": round(sum(times), 3), "block_ms_median": round(elapsed * 1e3, 3),
                       "block_ms_min": round(times[0] * 1e3, 3), "block_ms_max": round(times[-1] * 1e3, 3), "statistic": "median block"},
            "launch": {"mode": os.environ.get("GP_BENCH_LAUNCH", "direct" if world == 1 else "torch.distributed.run"),
                       "world_size_observed": (dist.get_world_size() if dist is not None else 1), "backend": backend},
            "whole_path_tflops": round(value * flop_per_pose / 1e12, 2), "roofline": None, "cpu_baseline": None, "single_sequence": single}), flush=True)


def single_sequence_trained(torch, dev, K, T0, n_obj):
    """One tracking sequence frame by frame with the TRAINED score / energy checkpoints (tests/golden/trained, scratch/train_synth.py) on a
    synthetic sequence with known poses (synth.posed_sequence): what a frame costs when the network actually pulls the candidates into a mode -
    the seeded random network of the main legs wanders, and its adaptive solve takes 4-8 attempts per frame instead of ~4."""
    import numpy as np
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.runner import TrackingRunner
    ck = {m: os.path.join(ROOT, "tests", "golden", "trained", f"ckpt_{m}.pth") for m in ("score", "energy")}
    sa = PoseNet(get_config(device=dev, posenet_mode="score", sampler_mode=["ode"]))
    ea = PoseNet(get_config(device=dev, posenet_mode="energy"))
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):  # (load_ckpt prints the path, like the reference; the line must stay the only stdout)
        sa.load_ckpt(model_dir=ck["score"], model_path=True, load_model_only=True)
        ea.load_ckpt(model_dir=ck["energy"], model_path=True, load_model_only=True)
    nfr = 30
    seq = synth.posed_sequence(11, n_frames=nfr, n_obj=n_obj)
    frames = [torch.from_numpy(seq["pts"][f]).to(dev) for f in range(nfr)]
    gt = torch.eye(4).repeat(n_obj, 1, 1)
    gt[:, :3, :3], gt[:, :3, 3] = torch.from_numpy(seq["R"][0]).float(), torch.from_numpy(seq["t"][0]).float()
    names = [f"o{j}" for j in range(n_obj)]
    tr = TrackingRunner(sa, ea, repeat_num=K, T0=T0)
    for f in range(8):
        tr.step(frames[f], names, gt)
    torch.cuda.synchronize()
    per, nfe, terr = [], [], []
    for rep in range(7):
        tr.reset()  # every repeat is the 30-frame sequence from its first frame
        t0 = time.perf_counter()
        evals = 0
        for f in range(nfr):
            out = tr.step(frames[f], names, gt)
            evals += int(sa.net.last_sampler.last_stats["nfev"])
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / nfr)
        nfe.append(evals / nfr)
        terr.append(float(np.median(np.linalg.norm(out["average_sRT"][:, :3, 3].cpu().numpy() - seq["t"][nfr - 1], axis=1)) * 100))
    dt = statistics.median(per)
    return {"ms_per_frame": round(dt * 1e3, 3), "ms_per_frame_min": round(min(per) * 1e3, 3), "ms_per_frame_max": round(max(per) * 1e3, 3), "repeats": len(per),
            "frames_per_repeat": nfr, "statistic": "median of the repeats", "mean_nfev_per_frame_by_repeat": [round(v, 1) for v in nfe],
            "median_translation_error_cm_at_the_last_frame": round(statistics.median(terr), 2), "objects_per_frame": n_obj,
            "weights": "trained on synthetic posed clouds (tests/golden/trained)", "workload": "synth.posed_sequence(11): 30 frames, one TrackingRunner.step per frame"}


def pc_roofline(torch, smp, rows, n, flop_row=FLOP_SCORE_ROW):
    """`achieved` = algorithmic FLOPs of one FULL pc_step launch / its average duration.  The sampler graph is exactly n full
    launches + the short finish-only launch; both are timed with HIP events on the launch stream with nothing else in flight and
    the finish-only launch is subtracted, so the average is over full launches only."""
    reps = 5
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        smp.graph.replay()  # n + 1 pc_step launches, nothing else
    e1.record()
    torch.cuda.synchronize()
    chain_s = e0.elapsed_time(e1) * 1e-3 / reps
    nfin = 50
    smp.launch_step(n)
    torch.cuda.synchronize()
    gfin = torch.cuda.CUDAGraph()  # captured, so that the host's launch rate does not show up in a 5 us kernel
    with torch.cuda.graph(gfin):
        for _ in range(nfin):
            smp.launch_step(n)  # finish-only launch (no score evaluation)
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gfin.replay()
    f0.record()
    gfin.replay()
    f1.record()
    torch.cuda.synchronize()
    fin_s = min(f0.elapsed_time(f1) * 1e-3 / nfin, chain_s / (n + 1))
    per_launch_s = (chain_s - fin_s) / n
    flops_per_launch = rows * flop_row
    ach = flops_per_launch / per_launch_s / 1e12
    roofline = {"bound": "mfma", "kernel": smp.kernel_name, "rows_per_launch": rows, "achieved": round(ach, 2),
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                "avg_launch_us": round(per_launch_s * 1e6, 2), "finish_launch_us": round(fin_s * 1e6, 2), "full_launches_per_chain": n,
                "flops_per_launch": flops_per_launch}
    # HBM-side bytes per launch come from the PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE in separate
    # rocprofv3 runs, gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md); only valid for the profiled shape
    for name in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(tpath):
            continue
        try:
            tj = json.load(open(tpath))[f"{roofline['kernel']}@{rows}"]
            roofline["traffic"] = tj["corrected_bytes_per_launch"]
            roofline["traffic_note"] = (f"PMC, profiles/{name}: raw {tj['raw_bytes_per_launch']} B, algorithmic "
                                        f"{tj['algorithmic_bytes_per_launch']} B; Infinity-Cache hits are counted (the 1 MB weight set "
                                        "is re-fetched by each of the 8 XCD L2s every launch)")
            break
        except (KeyError, ValueError):
            pass
    return roofline


def one_batch_leg(torch, score_agent, pool, B, K, n):
    from genpose_amd.pipeline import PipelinedPCPredictor
    p1 = PipelinedPCPredictor(score_agent, B, K, n, batches_per_launch=1, overlap=False)
    p1.run(pool[:3])
    torch.cuda.synchronize()
    nb = 40
    t1 = time.perf_counter()
    p1.run([pool[j % len(pool)] for j in range(nb)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    r = pc_roofline(torch, p1._sampler(0, 1), B * K, n)
    out = {"value": round(B * nb / dt, 2), "unit": "poses/s", "ms_per_step": round(dt / nb * 1e3, 3), "kernel": r["kernel"],
           "rows_per_launch": B * K, "avg_launch_us": r["avg_launch_us"], "frac": r["frac"]}
    # (two launch chains in flight - `--sampler-streams 2` - were re-measured in round 6: 21 k poses/s in a quiet process, 13-20 k inside this one:
    #  bimodal, not reported; EXPERIMENTS.md section H.6)
    return out


def split_bf16_leg(torch, score_agent, pool, B, K, n, G, dev, sampler_too):
    """OPT-IN, EXPLORATORY (round 5; csrc/sa_bf16x3.hip, csrc/trunk_bf16x3.hip): the headline workload with dense layers on the bf16 matrix
    pipe as three-term split products (a.b ~= a_hi.b_hi + a_lo.b_hi + a_hi.b_lo, fp32 accumulate) - grouping levels 1-2 of the encoder
    (`encoder_precision`), and with `sampler_too` also the three dense layers of the PC sampler's score network (`sampler_precision`).
    SEPARATE legs: the headline `value` above is pure fp32.  Reported with the measured deviation: encoder features against the fp32
    kernels' on the same clouds, and - same prior and noise draws through both paths - the PC-100 poses against the fp32 path's."""
    from genpose_amd.config import get_config
    from genpose_amd.pipeline import PipelinedPCPredictor
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.weights_synth import make_state_dict
    agent = PoseNet(get_config(device=dev, posenet_mode="score", sampler_mode=["pc"], sampling_steps=n, encoder_precision="bf16x3",
                               sampler_precision="bf16x3" if sampler_too else "f32"))
    agent.load_state_dict(make_state_dict(0, "score"))
    pipe = PipelinedPCPredictor(agent, B, K, n, batches_per_launch=G, overlap=False)  # one stream, like the headline run
    batches = lambda count: [pool[j % len(pool)] for j in range(count)]
    pipe.run(batches(G))
    pipe.run(batches(G))
    torch.cuda.synchronize()
    pipe.timing = True
    times = []
    nb = 2 * G
    for _ in range(7):
        t0 = time.perf_counter()
        for _o in pipe.run(batches(nb)):
            pass
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    in_situ = pipe.sampler_launch_seconds()
    out = {"value": round(B * nb / dt, 2), "unit": "poses/s", "ms_per_step": round(dt / nb * 1e3, 3), "batches_per_launch": G,
           "sampler_kernel": pipe._sampler(0, G).kernel_name, "sampler_in_situ_avg_launch_us": round(in_situ * 1e6, 2) if in_situ else None,
           "dtype": "f32 via 3 x bf16 split products, fp32 accumulate: encoder grouping levels 1-2" + (" and the PC sampler's score network" if sampler_too else "")
                    + "; everything else f32",
           "opt_in": "cfg.encoder_precision = 'bf16x3'" + (", cfg.sampler_precision = 'bf16x3'" if sampler_too else "") + " (defaults 'f32')"}
    big = torch.cat(batches(G), dim=0)
    if not sampler_too:
        f32 = score_agent.net.pts_encoder.forward(big).clone()
        fbf = agent.net.pts_encoder.forward(big).clone()
        scale = float(f32.abs().max())
        d = (fbf - f32).abs()
        reps = 10
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for enc_, a, b in ((score_agent.net.pts_encoder, ev[0], ev[1]), (agent.net.pts_encoder, ev[2], ev[3])):
            for _ in range(3):
                enc_.encode(big)
            a.record()
            for _ in range(reps):
                enc_.encode(big)
            b.record()
        torch.cuda.synchronize()
        out["encoder_pass_ms"] = {"clouds": int(big.shape[0]), "f32": round(ev[0].elapsed_time(ev[1]) / reps, 3), "bf16x3": round(ev[2].elapsed_time(ev[3]) / reps, 3)}
        out["feature_deviation_vs_f32"] = {"max_abs_over_scale": float(d.max()) / scale, "rms_over_scale": float(d.pow(2).mean().sqrt()) / scale,
                                           "clouds": int(big.shape[0]), "note": "per-level error against fp64: tests/test_gpu_bf16x3.py, profiles/r5_bf16x3_gate.txt"}
    else:
        # the same draws through the fp32 path and this one: how far the PC-100 poses move
        gen = torch.Generator().manual_seed(5)
        R1 = B * K
        priors = [torch.randn(R1, 9, generator=gen).to(dev) for _ in range(G)]
        noises = [(torch.randn(n, R1, 9, generator=gen).to(dev), torch.randn(n, R1, 9, generator=gen).to(dev)) for _ in range(G)]
        ref_pipe = PipelinedPCPredictor(score_agent, B, K, n, batches_per_launch=G, overlap=False)
        ref = torch.stack([o.clone() for o in ref_pipe.run(batches(G), prior_noise=priors, noise=noises)])
        got = torch.stack([o.clone() for o in pipe.run(batches(G), prior_noise=priors, noise=noises)])
        torch.cuda.synchronize()
        rot = (got[..., :6] - ref[..., :6]).abs().reshape(-1)
        tr = (got[..., 6:] - ref[..., 6:]).abs().max() / ref[..., 6:].abs().max()
        rq = torch.quantile(rot[: 1 << 24].float(), torch.tensor([0.5, 0.99, 0.999], device=rot.device))
        out["pose_deviation_vs_f32_same_draws"] = {"rotation_abs_median": float(rq[0]), "rotation_abs_p99": float(rq[1]), "rotation_abs_p999": float(rq[2]),
                                                   "rotation_abs_max": float(rot.max()),
                                                   "translation_max_over_scale": float(tr), "clouds": int(G * B),
                                                   "note": "the 100-step recursion renormalises the rotation columns every step and amplifies a perturbation for the rare row whose "
                                                           "column passes near zero: two fp32 launch plans of one batch differ by p99.9 2.6e-5 / max 7e-3 "
                                                           "(tests/test_gpu_fullsize.py), this arithmetic (8e-6 of the score's scale per evaluation against fp64, "
                                                           "tests/test_gpu_bf16x3.py) by more - every draw is a valid sample of the same sampler, but NOT "
                                                           "within the parity tolerance: opt-in only"}
    return out


def ode_leg(torch, B, K, G, T0, pool, dev):
    """Secondary mode ODE-100: cond_ode_sampler semantics, sampling_steps=100, T0=0.55, adaptive RK45 on the device."""
    from genpose_amd.config import get_config
    from genpose_amd.pipeline import GroupedODEPredictor
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.weights_synth import make_state_dict
    agent = PoseNet(get_config(device=dev, posenet_mode="score", sampler_mode=["ode"], sampling_steps=100))
    agent.load_state_dict(make_state_dict(0, "score"))
    pred = GroupedODEPredictor(agent, B, K, T0=T0, batches_per_launch=G)
    pred.run([pool[j % len(pool)] for j in range(G)])
    torch.cuda.synchronize()
    nb = 4 * G
    t0 = time.perf_counter()
    pred.run([pool[j % len(pool)] for j in range(nb)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nfev = int(round(sum(pred.last_nfev) / max(1, len(pred.last_nfev))))
    return {"value": round(B * nb / dt, 2), "unit": "poses/s", "ms_per_step": round(dt / nb * 1e3, 3), "nfev": nfev, "batches_per_launch": G,
            "workload": f"{B} clouds x {K} cand, cond_ode_sampler(sampling_steps=100, T0={T0}), RK45 rtol=atol=1e-5 on the device"}


def full_pipeline_leg(torch, K, n, dev, B=256):
    """BASELINE configs[2]: score + energy agents, PC-100 sampler, energy ranking, top-60 % aggregation at 256 clouds."""
    from genpose_amd import synth
    from genpose_amd.config import get_config
    from genpose_amd.pipeline import FullPipelinePredictor
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.weights_synth import make_state_dict
    sa = PoseNet(get_config(device=dev, posenet_mode="score", sampler_mode=["pc"], sampling_steps=n))
    sa.load_state_dict(make_state_dict(0, "score"))
    ea = PoseNet(get_config(device=dev, posenet_mode="energy"))
    ea.load_state_dict(make_state_dict(0, "energy"))
    G = 5  # batches of 256 clouds per launch chain (each with its own batch-global coupling): the shape tests/test_gpu_fullsize.py checks
    pool = [torch.from_numpy(synth.make_batch(B, start=4096 + B * j)).to(dev) for j in range(G)]  # distinct clouds in every batch of a launch
    pts = pool[0]
    fp = FullPipelinePredictor(sa, ea, B, K, n, batches_per_launch=G)
    fp.run_many(pool)
    fp.run_many(pool)
    torch.cuda.synchronize()
    nb = 3 * G
    t0 = time.perf_counter()
    fp.run_many([pool[j % G] for j in range(nb)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for _ in range(2):
        fp.run(pts)  # builds / captures the one-batch sampler outside the timing
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for j in range(6):
        fp.run(pool[j % G])
    torch.cuda.synchronize()
    dt1 = (time.perf_counter() - t1) / 6
    return {"value": round(B * nb / dt, 2), "unit": "poses/s", "ms_per_step": round(dt / nb * 1e3, 3), "clouds": B, "batches_per_launch": G,
            "one_batch_per_launch": {"value": round(B / dt1, 2), "ms_per_step": round(dt1 * 1e3, 3)},
            "workload": f"{B} clouds x {K} cand: encoder + PC-{n} sampler (score model) | encoder + energy (energy model) -> ranking -> top-60% aggregate",
            "flop_per_pose": 7.10e9, "whole_path_tflops": round(B * nb / dt * 7.10e9 / 1e12, 2),
            "executed_tflops": round(B * nb / dt * (7.10e9 - 2 * (FLOP_ENCODER - FLOP_ENCODER_EXECUTED)) / 1e12, 2)}


class _HostSyncCounter:
    """Counts the host-side waits for the device inside a block: torch.cuda.synchronize, Stream / Event.synchronize and the blocking
    device-to-host reads (.cpu(), .item(), .tolist(), .numpy() of a device tensor)."""

    def __init__(self, torch):
        self.torch, self.n, self.saved = torch, 0, []

    def _wrap(self, owner, name, only_cuda_tensor=False):
        orig = getattr(owner, name)
        counter = self

        def wrapped(*a, **kw):
            if not only_cuda_tensor or (a and getattr(a[0], "is_cuda", False)):
                counter.n += 1
            return orig(*a, **kw)
        self.saved.append((owner, name, orig))
        setattr(owner, name, wrapped)

    def __enter__(self):
        t = self.torch
        self._wrap(t.cuda, "synchronize")
        self._wrap(t.cuda.Stream, "synchronize")
        self._wrap(t.cuda.Event, "synchronize")
        for name in ("cpu", "item", "tolist", "numpy"):
            self._wrap(t.Tensor, name, only_cuda_tensor=True)
        return self

    def __exit__(self, *exc):
        for owner, name, orig in reversed(self.saved):
            setattr(owner, name, orig)


def drop_in_leg(torch, dev, K, B=256, T0=0.55):
    """The call sequence a GenPose user gets from the one-line import swap of INTEGRATION.md §2, measured through the AGENT API only -
    scripts/eval_single.sh:1-16 + runners/evaluation_single.py:309-353,356-489: batches of 256 instances, K = 50 candidates,
        pred = score_agent.pred_func(data, repeat_num=K, T0=0.55)         # ODE sampler (the script's default), adaptive RK45
        energy = energy_agent.get_energy(data, pose_samples=pred, T=1e-5)
        rank + top-60 % aggregation
    one batch per call, nothing shared between calls, no request batching, no predictor object.  Also with the PC-100 sampler (the
    benched sampler) and, beside it, FullPipelinePredictor.run on the same batches (one batch per launch): what the agent path gives up
    against the predictor that overlaps the energy encoder with the sampler."""
    from genpose_amd import reward, synth
    from genpose_amd.config import get_config
    from genpose_amd.pipeline import FullPipelinePredictor
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.runner import make_batch_sample
    from genpose_amd.weights_synth import make_state_dict
    pool = [torch.from_numpy(synth.make_batch(B, start=20000 + B * j)).to(dev) for j in range(4)]
    ea = PoseNet(get_config(device=dev, posenet_mode="energy"))
    ea.load_state_dict(make_state_dict(0, "energy"))
    out = {"workload": f"{B} clouds x {K} cand per call: PoseNet.pred_func -> PoseNet.get_energy -> rank_aggregate(ratio=0.6), agent API only "
                       "(scripts/eval_single.sh, evaluation_single.py:356-489)"}
    nb = 12
    for name, sampler, steps in (("ode_T0_0.55", "ode", None), ("pc_100", "pc", 100)):
        sa = PoseNet(get_config(device=dev, posenet_mode="score", sampler_mode=[sampler], sampling_steps=steps))
        sa.load_state_dict(make_state_dict(0, "score"))

        def call(pts):
            data = make_batch_sample(pts)  # evaluation_single.py:394-403
            pred = sa.pred_func(data=data, repeat_num=K, save_path=None, T0=T0)
            energy = ea.get_energy(data=data, pose_samples=pred, T=1e-5)
            return reward.rank_aggregate(pred, energy, ratio=0.6)["avg_pose"]

        for j in range(4):
            call(pool[j % len(pool)])  # builds the samplers, captures the graphs (encoder passes from the second call on)
        torch.cuda.synchronize()
        with _HostSyncCounter(torch) as hs:
            t0 = time.perf_counter()
            for j in range(nb):
                call(pool[j % len(pool)])
            issued = time.perf_counter() - t0
            nsync = hs.n
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        leg = {"value": round(B * nb / dt, 2), "unit": "poses/s", "ms_per_call": round(dt / nb * 1e3, 3), "host_syncs_per_call": round(nsync / nb, 2),
               "host_issue_ms_per_call": round(issued / nb * 1e3, 3), "sampler": sampler}
        if sampler == "ode":
            st = sa.net.last_sampler.last_stats
            leg["nfev"], leg["attempts"] = int(st["nfev"]), int(st["n_attempts"])
            leg["replays"] = dict(sa.net.last_sampler.last_replays)
        else:
            fp = FullPipelinePredictor(sa, ea, B, K, steps)
            for j in range(3):
                fp.run(pool[j % len(pool)])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for j in range(nb):
                fp.run(pool[j % len(pool)])
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            leg["predictor_one_batch_per_launch"] = {"value": round(B * nb / d1, 2), "ms_per_call": round(d1 / nb * 1e3, 3)}
            leg["agent_over_predictor"] = round(d1 / dt, 4)
        out[name] = leg
    try:  # the same ODE call sequence on the TRAINED checkpoints and held-out posed clouds (informative; never costs the line)
        import contextlib
        import io
        ck = {m: os.path.join(ROOT, "tests", "golden", "trained", f"ckpt_{m}.pth") for m in ("score", "energy")}
        sa = PoseNet(get_config(device=dev, posenet_mode="score", sampler_mode=["ode"]))
        et = PoseNet(get_config(device=dev, posenet_mode="energy"))
        with contextlib.redirect_stdout(io.StringIO()):
            sa.load_ckpt(model_dir=ck["score"], model_path=True, load_model_only=True)
            et.load_ckpt(model_dir=ck["energy"], model_path=True, load_model_only=True)
        tpool = [torch.from_numpy(synth.posed_batch(range(3_000_000 + B * j, 3_000_000 + B * (j + 1)))["pts"]).to(dev) for j in range(2)]

        def tcall(pts):
            data = make_batch_sample(pts)
            pred = sa.pred_func(data=data, repeat_num=K, save_path=None, T0=T0)
            return reward.rank_aggregate(pred, et.get_energy(data=data, pose_samples=pred, T=1e-5), ratio=0.6)["avg_pose"]
        for j in range(4):
            tcall(tpool[j % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(nb):
            tcall(tpool[j % 2])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = sa.net.last_sampler.last_stats
        out["ode_T0_0.55_trained_weights"] = {"value": round(B * nb / dt, 2), "unit": "poses/s", "ms_per_call": round(dt / nb * 1e3, 3), "nfev": int(st["nfev"]),
                                              "attempts": int(st["n_attempts"]), "weights": "trained on synthetic posed clouds (tests/golden/trained)",
                                              "clouds": "held-out synth.make_posed_cloud instances"}
    except Exception as exc:  # noqa: BLE001
        out["ode_T0_0.55_trained_weights"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def config0_leg(torch, dev):
    """BASELINE configs[0] - the reference's own CPU-runnable case: ONE cloud x 1024 pts, 10 candidates, 20 SDE steps (PC) and the default
    adaptive ODE solve (T0 = 0.55), evaluation_single.py call sequence through the agent API: latency per call on the device (the work of a
    call is far too small to fill the chip: this is a latency figure, not a throughput one).  The oracle's time for the same call on the
    host sits in `cpu_baseline.config0_single_object`."""
    from genpose_amd import reward, synth
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.runner import make_batch_sample
    from genpose_amd.weights_synth import make_state_dict
    B, K = 1, 10
    pool = [torch.from_numpy(synth.make_batch(B, start=30000 + j)).to(dev) for j in range(4)]
    ea = PoseNet(get_config(device=dev, posenet_mode="energy"))
    ea.load_state_dict(make_state_dict(0, "energy"))
    out = {"workload": "1 cloud x 1024 pts, 10 candidates per call: pred_func -> get_energy -> rank_aggregate (agent API)"}
    for name, sampler, steps in (("pc_20", "pc", 20), ("ode_T0_0.55", "ode", None)):
        sa = PoseNet(get_config(device=dev, posenet_mode="score", sampler_mode=[sampler], sampling_steps=steps))
        sa.load_state_dict(make_state_dict(0, "score"))

        def call(pts):
            data = make_batch_sample(pts)
            pred = sa.pred_func(data=data, repeat_num=K, save_path=None, T0=0.55)
            energy = ea.get_energy(data=data, pose_samples=pred, T=1e-5)
            return reward.rank_aggregate(pred, energy, ratio=0.6)["avg_pose"]

        for j in range(6):
            call(pool[j % 4])
        torch.cuda.synchronize()
        nb, reps = 30, 7
        per = []
        for _ in range(reps):  # a latency figure moves by ~20 % between sessions and by several % inside one: median AND range are reported
            t0 = time.perf_counter()
            for j in range(nb):
                call(pool[j % 4])
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / nb)
        dt = statistics.median(per)
        out[name] = {"ms_per_call": round(dt * 1e3, 3), "ms_per_call_min": round(min(per) * 1e3, 3), "ms_per_call_max": round(max(per) * 1e3, 3),
                     "repeats": reps, "calls_per_repeat": nb, "statistic": "median of the repeats", "ms_per_call_by_repeat": [round(v * 1e3, 3) for v in per],
                     "poses_per_s": round(B / dt, 1)}
        if sampler == "ode":
            out[name]["nfev"] = int(sa.net.last_sampler.last_stats["nfev"])
    return out


def energy_model_leg(torch, dev, B, K, G, n=20):
    """Roofline of the forward + vector-Jacobian right-hand side: the PC step of a sampler that draws FROM the energy model (its score is
    the gradient of the inner-product energy, energynet.py:200-222: forward pass + backward pass per evaluation, 2 x 0.5335 MFLOP per row
    on the minimal count) at the benched launch size (G x B clouds x K candidates), chain form (csrc/trunk_chain_vjp.h), and the 16-row
    tile form (csrc/score_bwd.h) beside it."""
    from genpose_amd.samplers import PCSampler
    from genpose_amd.scorenet import ScoreNetHIP
    from genpose_amd.weights_synth import make_state_dict
    net = ScoreNetHIP(make_state_dict(0, "energy"), dev)
    out = {}
    for tile in (0, 16):
        smp = PCSampler(net, G * B, K, n, dev, groups=G, model="energy", tile=tile)
        cvec, centre = torch.randn(G * B, 768, device=dev), torch.randn(G * B, 3, device=dev)
        x0 = torch.randn(G * B * K, 9, device=dev) * 50.0
        for _ in range(2):
            smp.run(cvec, centre, x0)
        r = pc_roofline(torch, smp, G * B * K, n, flop_row=2 * FLOP_SCORE_ROW)
        out["plan" if tile == 0 else "tile16"] = {k: r[k] for k in ("kernel", "rows_per_launch", "avg_launch_us", "achieved", "frac", "flops_per_launch")}
    return out


def run_cpu_baseline(torch, args, K, n):
    """The oracle (CPU restatement of the reference path, validated against the imported reference) timed on the host cores of this
    box on a bounded sample of the same workload: ONE batch of `--cpu-clouds` (64) clouds, as BASELINE.md §3 plans.  Checker
    infrastructure used as a baseline: allowed use of oracle/ (task statement §3).  The two stages are swept over thread counts
    SEPARATELY (the encoder is conv-like work over 12 MB of grouped rows per cloud, the sampler 100 small GEMMs over 3200 rows: their
    optima differ), and the end-to-end sample (encoder + sampler, what `value` reports) runs each stage at ITS best count."""
    from genpose_amd import synth
    from oracle import genpose_oracle as go
    from oracle import parallel as opar
    from oracle import pn2_oracle as ops
    Bc = args.cpu_clouds
    budget = args.cpu_budget
    t_start = time.perf_counter()
    sd = go.make_state_dict(0, "score")
    pts = torch.from_numpy(synth.make_batch(Bc, start=0))
    cen = pts.mean(dim=1)
    gen = torch.Generator().manual_seed(0)
    prior = torch.randn(Bc * K, 9, generator=gen)
    z1 = z2 = None
    if args.sampler == "pc":
        z1, z2 = torch.randn(n, Bc * K, 9, generator=gen), torch.randn(n, Bc * K, 9, generator=gen)
    ncores = os.cpu_count() or 1
    cands = [t for t in (8, 16, 32, 64, 128, 256) if t <= ncores] or [ncores]
    saved = torch.get_num_threads()
    cen_r = cen.repeat_interleave(K, 0)

    def set_threads(t):
        torch.set_num_threads(t)
        ops.opt_n_threads(t)

    def encoder(threads):
        set_threads(threads)
        return go.encoder_forward(sd, pts)

    def sampler(threads, feat):
        set_threads(threads)
        feat_r = feat.repeat_interleave(K, 0)
        fn = lambda x, t: go.score_forward(sd, feat_r, x, t)
        if args.sampler == "pc":
            go.pc_sampler(fn, prior * float(go.ve_sigma(1.0)), cen_r, n, z1, z2)
        else:
            go.ode_sampler(fn, prior * float(go.ve_sigma(torch.tensor(0.55))), cen_r, 0.55)

    def sweep(fn, share):
        """Seconds per thread count, ascending; stops once a count is 1.3x slower than the best so far or the stage's share of the
        budget is spent."""
        t_in, out = time.perf_counter(), {}
        for t in cands:
            t0 = time.perf_counter()
            fn(t)
            out[t] = time.perf_counter() - t0
            if out[t] > 1.3 * min(out.values()) or time.perf_counter() - t_in > share * budget:
                break
        return out

    try:
        set_threads(min(16, ncores))
        go.pred_func(sd, pts[:2], cen[:2], 2, "pc", prior[:4], sampling_steps=2, z_langevin=torch.zeros(2, 4, 9), z_predictor=torch.zeros(2, 4, 9))  # library init
        feat = encoder(min(16, ncores))  # warm (page-in, thread pools); its features feed the sampler sweep
        enc_s = sweep(encoder, 0.3)
        smp_s = sweep(lambda t: sampler(t, feat), 0.3)
        best_enc, best_smp = min(enc_s, key=enc_s.get), min(smp_s, key=smp_s.get)
        # clouds are independent units: the encoder also runs as slices of 16 clouds in worker PROCESSES side by side (oracle/parallel.py:
        # one process with many threads does not scale on its small convolutions - 64 threads are slower than 32 - several processes
        # with 8 threads each do); taken when it is faster.  The sampler's rows are coupled through the batch-mean score norm every
        # step: one process.
        pool_workers, pool_threads = opar._plan()
        enc_pool_s, pool_same = None, None
        if pool_workers > 1 and Bc >= 32:
            opar.encoder_features("score", synth.make_batch(32, start=40000))  # starts the workers (not timed)
            opar._cache.clear()
            t0 = time.perf_counter()
            feat_pool = torch.from_numpy(opar.encoder_features("score", pts))
            enc_pool_s = time.perf_counter() - t0
            pool_same = bool(torch.equal(feat_pool, feat))  # the same function on the same clouds
        use_pool = enc_pool_s is not None and enc_pool_s < min(enc_s.values())

        def encoder_best():
            if not use_pool:
                return encoder(best_enc)
            opar._cache.clear()  # (the cache is for the tests: every timed run computes)
            return torch.from_numpy(opar.encoder_features("score", pts))
        runs = []
        while True:
            t0 = time.perf_counter()
            sampler(best_smp, encoder_best())
            runs.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start + runs[-1] > budget or len(runs) >= 10:
                break
        # BASELINE configs[0] (one cloud, 10 candidates, 20 PC steps / the default ODE solve + energies) on the host: the oracle's latency
        # per call beside the device's `config0_single_object` figures (a fraction of a second of CPU work)
        set_threads(min(8, ncores))
        p1 = torch.from_numpy(synth.make_batch(1, start=30000))
        sde_ = go.make_state_dict(0, "energy")
        c0 = {"threads": min(8, ncores)}
        for name, smp, kw in (("pc_20", "pc", dict(sampling_steps=20, z_langevin=torch.randn(20, 10, 9, generator=gen), z_predictor=torch.randn(20, 10, 9, generator=gen))),
                              ("ode_T0_0.55", "ode", dict(T0=0.55))):
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                pred1, _, _ = go.pred_func(sd, p1, p1.mean(dim=1), 10, smp, prior[:10], **kw)
                go.get_energy(sde_, p1, p1.mean(dim=1), pred1, T=1e-5)
                ts.append(time.perf_counter() - t0)
            c0[name + "_ms_per_call"] = round(statistics.median(ts) * 1e3, 1)
    finally:
        set_threads(saved)
        opar.shutdown()
    med = statistics.median(runs)
    enc_cores = min(pool_workers, (Bc + 15) // 16) * pool_threads if use_pool else best_enc
    return {"value": round(Bc / med, 3), "unit": "poses/s", "cores": max(enc_cores, best_smp), "kind": "port", "host_cores": ncores,
            "config0_single_object": c0,
            "threads": {"encoder": best_enc, "sampler": best_smp},
            "encoder_process_pool": {"used": bool(use_pool), "seconds": None if enc_pool_s is None else round(enc_pool_s, 3),
                                     "workers_x_threads": [min(pool_workers, (Bc + 15) // 16), pool_threads],
                                     "features_equal_one_process": pool_same},
            "encoder_s_by_threads": {t: round(v, 3) for t, v in enc_s.items()}, "sampler_s_by_threads": {t: round(v, 3) for t, v in smp_s.items()},
            "end_to_end_runs_s": [round(r, 2) for r in runs],
            "sample": f"{len(runs)} x ({Bc} clouds x 1024 pts, {K} cand, {args.sampler.upper()} {n} steps), encoder + sampler end to end, median; "
                      f"each stage at its own best configuration (encoder {'%d processes x %d threads (slices of 16 clouds)' % (min(pool_workers, (Bc + 15) // 16), pool_threads) if use_pool else '%d threads' % best_enc}, "
                      f"sampler {best_smp} threads; swept {sorted(set(enc_s) | set(smp_s))}); "
                      f"{time.perf_counter() - t_start:.1f} s of CPU work in total; oracle/genpose_oracle.py (torch-CPU fp32 MLPs + OpenMP C ops)"}


if __name__ == "__main__":
    main()
