"""bench.py launch mechanics on the GPU box: `python bench.py --gpus N` must run without an external launcher (it spawns
`python -m torch.distributed.run` itself), rank 0 prints exactly one JSON line, and the N-rank path is exercised on ONE device
(GP_BENCH_ONE_DEVICE=1: every rank uses cuda:0, collectives on gloo) so a 1-GPU box covers it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--repeats", "1", "--batch", "8", "--cand", "10", "--sde-steps", "6", "--batches-per-launch", "1",
         "--no-cpu-baseline", "--no-secondary"]
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline", "timing", "launch"}


def _run(extra, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + SMALL, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_self_launched_on_one_device():
    line = _run(["--gpus", "2"], {"GP_BENCH_ONE_DEVICE": "1"})
    assert REQUIRED <= set(line)
    assert line["n_gpus"] == 2 and line["launch"] == {"mode": "self", "world_size_observed": 2, "backend": "gloo"}
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["clouds_per_gpu"] == 8 and line["config"]["parallelism"] == "clouds sharded x2"
    assert abs(line["value"] - 2 * 8 * 2 / (line["ms_per_step"] * 2e-3)) <= 0.01 * line["value"]  # whole-job aggregate over both ranks


def test_one_rank_direct_and_through_the_launcher_agree():
    direct = _run(["--gpus", "1"])
    spawned = _run(["--gpus", "1"], {"GP_BENCH_FORCE_LAUNCH": "1"})
    assert direct["launch"]["mode"] == "direct" and spawned["launch"]["mode"] == "self"
    assert set(direct) == set(spawned) and REQUIRED <= set(direct)
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "data", "config", "higher_is_better"):
        assert direct[k] == spawned[k], k
    assert direct["roofline"]["kernel"] == spawned["roofline"]["kernel"]
    assert direct["roofline"]["flops_per_launch"] == spawned["roofline"]["flops_per_launch"]


def test_config3_full_pipeline_two_ranks_on_one_device():
    """BASELINE configs[3] shape (--pipeline full, clouds sharded, one gather of the aggregated poses per batch) through the N-rank path."""
    line = _run(["--gpus", "2", "--pipeline", "full"], {"GP_BENCH_ONE_DEVICE": "1"})
    assert REQUIRED <= set(line) and line["n_gpus"] == 2 and line["launch"]["world_size_observed"] == 2
    assert line["config"]["pipeline"] == "full" and "EnergyNet ranking" in line["config"]["workload"] and line["value"] > 0


def test_config4_tracking_two_ranks_on_one_device():
    """BASELINE configs[4]: --tracking (whole sequences per rank, MultiSequenceTracker per rank, results gathered per block)."""
    env = {"GP_BENCH_ONE_DEVICE": "1"}
    extra = ["--gpus", "2", "--tracking", "--sequences", "3", "--objects", "2", "--cand", "10", "--steps", "3", "--warmup", "2", "--repeats", "1"]
    envd = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        envd.pop(k, None)
    envd.update(env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=envd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert REQUIRED <= set(line) and line["n_gpus"] == 2 and line["config"]["sequences_per_gpu"] == 3 and line["config"]["sampler"] == "ode"
    assert line["value"] > 0 and abs(line["value"] - 2 * 3 * 2 * 3 / (line["ms_per_step"] * 3e-3)) <= 0.01 * line["value"]


# ------------------------------------------------------------------------------------------------------------------------------------
# Rehearsal of the 8-rank job (BASELINE configs[3] / configs[4]): the box has one GPU, so the eight ranks share cuda:0 and the
# collectives run on gloo - everything else (launcher, rank environment, sharding, barriers, MAX-reduced timing, the gather of the
# results, rank 0's one JSON line) is the code path the driver's `--gpus 8` run takes on an 8-GPU node.
def _run_raw(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_eight_ranks_plain_workload_on_one_device():
    line = _run(["--gpus", "8"], {"GP_BENCH_ONE_DEVICE": "1"})
    assert REQUIRED <= set(line) and line["n_gpus"] == 8
    assert line["launch"] == {"mode": "self", "world_size_observed": 8, "backend": "gloo"}
    assert line["scaling"] == "weak" and line["config"]["clouds_per_gpu"] == 8 and line["config"]["parallelism"] == "clouds sharded x8"
    assert abs(line["value"] - 8 * 8 * 2 / (line["ms_per_step"] * 2e-3)) <= 0.01 * line["value"]  # whole-job aggregate over the eight ranks


def test_config3_eight_ranks_2048_clouds_full_pipeline_on_one_device():
    """BASELINE configs[3] at its real shape: 8 ranks x 256 clouds = 2048 clouds per step, 50 candidates x 100 PC steps, energy ranking
    and aggregation, one gather of the aggregated poses per batch."""
    line = _run_raw(["--gpus", "8", "--pipeline", "full", "--batch", "256", "--steps", "2", "--warmup", "1", "--repeats", "1",
                     "--batches-per-launch", "1", "--no-cpu-baseline", "--no-secondary"], {"GP_BENCH_ONE_DEVICE": "1"})
    assert REQUIRED <= set(line) and line["n_gpus"] == 8 and line["launch"]["world_size_observed"] == 8
    c = line["config"]
    assert c["pipeline"] == "full" and c["clouds_per_gpu"] == 256 and c["candidates"] == 50 and c["sde_steps"] == 100
    assert c["workload"].startswith("configs[3]: 2048 clouds sharded over 8 GPUs = 256 clouds/GPU") and "EnergyNet ranking" in c["workload"]
    assert abs(line["value"] - 8 * 256 * 2 / (line["ms_per_step"] * 2e-3)) <= 0.01 * line["value"]  # 2048 clouds per step


def test_config4_tracking_eight_ranks_on_one_device():
    line = _run_raw(["--gpus", "8", "--tracking", "--sequences", "2", "--objects", "2", "--cand", "10", "--steps", "3", "--warmup", "2", "--repeats", "1"],
                    {"GP_BENCH_ONE_DEVICE": "1"})
    assert REQUIRED <= set(line) and line["n_gpus"] == 8 and line["launch"]["world_size_observed"] == 8
    assert line["config"]["sequences_per_gpu"] == 2 and line["config"]["parallelism"] == "whole sequences per rank x8 (replicas only)"
    assert line["value"] > 0 and abs(line["value"] - 8 * 2 * 2 * 3 / (line["ms_per_step"] * 3e-3)) <= 0.01 * line["value"]


def test_one_rank_on_rccl_costs_nothing():
    """GP_BENCH_FORCE_DIST=1: the N = 1 workload with the process group up on RCCL (backend 'nccl') - barrier, all-gather of every result and
    the MAX all-reduce of the block time inside the timed region.  The distributed plumbing must not cost throughput: the line stays within
    3 % of the plain N = 1 line (measured: < 1 % in four sessions; the bound leaves room for run-to-run noise of two separate processes)."""
    common = ["--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-secondary"]
    plain = _run_raw(common)
    rccl = _run_raw(common, {"GP_BENCH_FORCE_DIST": "1"})
    assert plain["launch"]["backend"] is None and rccl["launch"] == {"mode": "direct", "world_size_observed": 1, "backend": "nccl"}
    assert plain["config"] == rccl["config"] and plain["n_gpus"] == rccl["n_gpus"] == 1
    ratio = rccl["value"] / plain["value"]
    print(f"one rank on RCCL: {rccl['value']:.0f} poses/s against {plain['value']:.0f} plain ({100 * (ratio - 1):+.2f} %)")
    assert ratio > 0.97, (rccl["value"], plain["value"])


@pytest.mark.parametrize("n", [2, 4])
def test_two_and_four_ranks_on_one_device(n):
    """The N = 2 and N = 4 points of the driver's scaling curve through the same rehearsal as N = 8."""
    line = _run(["--gpus", str(n)], {"GP_BENCH_ONE_DEVICE": "1"})
    assert REQUIRED <= set(line) and line["n_gpus"] == n and line["launch"]["world_size_observed"] == n and "error" not in line
    assert abs(line["value"] - n * 8 * 2 / (line["ms_per_step"] * 2e-3)) <= 0.01 * line["value"]


@pytest.mark.parametrize("launcher", ["self", "external"])
def test_a_rank_that_dies_before_the_rendezvous_still_yields_a_line(launcher):
    """The driver's SCALE run is the first time RCCL sees more than one rank: whatever fails there must come back as a parseable line with an
    "error" field, not as a hang into the driver's time limit.  Rehearsal: rank 1 of 2 exits before init_process_group (GP_BENCH_KILL_RANK);
    rank 0 sits in the rendezvous until the launcher terminates it or the start timeout fires - either way it prints the line.
    'external' = the driver's own launch line (python -m torch.distributed.run ... bench.py), 'self' = bench.py spawning it."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"GP_BENCH_ONE_DEVICE": "1", "GP_BENCH_KILL_RANK": "1", "GP_BENCH_START_TIMEOUT": "20"})
    bench = os.path.join(ROOT, "bench.py")
    if launcher == "self":
        cmd = [sys.executable, bench, "--gpus", "2"] + SMALL
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29617",
               bench, "--gpus", "2"] + SMALL
    import time
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert p.returncode != 0
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["value"] is None and line["n_gpus"] == 2 and "error" in line, line
    # rank 0's own line (not the self-launcher's fallback for a launcher that died silently): it names the backend it was bringing up (None
    # if the launcher's SIGTERM arrived before the process group was started) and what torch sees of the box
    assert line["launch"]["rank"] == 0 and line["launch"]["backend"] in ("gloo", None) and line["launch"]["world_size_env"] == 2, line
    assert took < 120, took  # seconds, not the driver's 1 800 s limit
    print(f"killed-rank rehearsal ({launcher}): line after {took:.0f} s: {line['error']}")


def test_more_ranks_than_devices_yields_a_line():
    """The real RCCL start path with a rank that cannot get a device (two ranks on this one-GPU box, no GP_BENCH_ONE_DEVICE): rank 1 reports
    'LOCAL_RANK 1 but torch sees 1 device(s)' and leaves; rank 0 is inside init_process_group('nccl') waiting for it when the launcher
    terminates it - and still prints its one line."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("written for a one-GPU box")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GP_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    env["GP_BENCH_START_TIMEOUT"] = "30"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["value"] is None and line["n_gpus"] == 2 and "error" in line and line["launch"]["rank"] == 0, line
    assert line["launch"]["backend"] in ("nccl", None) and line["launch"]["device_count"] == 1, line
    assert "LOCAL_RANK 1 but torch sees 1 device(s)" in p.stderr
