"""CPU, world_size 2, gloo: the sharding + single-all-gather logic of genpose_amd.dist (the N > 1 path of bench.py and
of the runners).  The per-rank compute is a stand-in here (the CPU oracle's encoder on a couple of clouds, or a cheap
function): what is under test is that shards, padding and gather order reproduce the unsharded result exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from genpose_amd import dist as gdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_cover_everything():
    for n in [0, 1, 2, 5, 64, 2048, 2049]:
        for world in [1, 2, 3, 8]:
            b = [gdist.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1


def _infer(clouds):
    # deterministic per-cloud function with several outputs / dtypes (like pred_pose f64, energy f32, avg f32)
    c = clouds.double().mean(dim=1)
    return {"pred_pose": torch.stack([c * (k + 1) for k in range(4)], dim=1).repeat(1, 1, 3),  # [n,4,9] f64
            "energy": clouds.float().std(dim=1)[:, None, :2].repeat(1, 4, 1),                  # [n,4,2] f32
            "avg": clouds.float().amax(dim=1)}                                                 # [n,3]


def _worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        clouds = torch.randn(n, 32, 3, generator=g)
        out = gdist.ShardedInference(_infer)(clouds)
        ref = _infer(clouds)
        ok = all(torch.equal(out[k], ref[k]) and out[k].dtype == ref[k].dtype for k in ref)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 8, 2])
def test_sharded_inference_world2(n):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _worker_oracle(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from genpose_amd import synth
        from oracle import genpose_oracle as go
        torch.set_num_threads(2)
        sd = go.make_state_dict(0, "score")
        clouds = torch.from_numpy(synth.make_batch(3, start=40))
        fn = lambda c: {"feat": go.encoder_forward(sd, c)}
        out = gdist.ShardedInference(fn)(clouds)
        ref = fn(clouds)
        ret[rank] = float((out["feat"] - ref["feat"]).abs().max())
    finally:
        dist.destroy_process_group()


def test_clouds_are_independent_units_world2():
    """The property sharding relies on: a cloud's encoder output does not depend on its batch-mates."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_oracle, args=(2, port, ret), nprocs=2, join=True)
    assert max(ret.values()) == 0.0


# ---------------------------------------------------------------------------------------------- sub-group source rank
def _worker_subgroup(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sub = dist.new_group(ranks=[1, 2])  # does NOT contain global rank 0: the broadcast source must be the group's rank 0
        if rank in (1, 2):
            g = torch.Generator().manual_seed(0)
            clouds = torch.randn(5, 16, 3, generator=g)
            out = gdist.ShardedInference(_infer, group=sub)(clouds)
            ref = _infer(clouds)
            ret[rank] = all(torch.equal(out[k], ref[k]) for k in ref)
        else:
            ret[rank] = True
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_inference_on_a_subgroup_without_global_rank0():
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_subgroup, args=(3, port, ret), nprocs=3, join=True)
    assert dict(ret) == {0: True, 1: True, 2: True}


# ---------------------------------------------------------------------------------------------- tracking dispatch (configs[4])
class _StubTracker:
    """Stands in for runner.MultiSequenceTracker on the CPU: a per-sequence state machine whose output depends on the frame AND on
    the sequence's previous output (the warm start), so a frame handled by the wrong rank / out of order changes the result."""

    def __init__(self, n):
        self.prev = [torch.zeros(4, 4, dtype=torch.float64) for _ in range(n)]
        self.calls = 0

    def step(self, frames):
        self.calls += 1
        out = []
        for i, f in enumerate(frames):
            if f is None:
                out.append(None)
                continue
            pts, names, gt = f
            cur = 0.5 * self.prev[i] + pts.double().mean(dim=(0, 1)).sum() * torch.eye(4, dtype=torch.float64) + gt[0].double()
            self.prev[i] = cur
            out.append({"average_sRT": cur.unsqueeze(0).repeat(pts.shape[0], 1, 1)})
        return out


def _frames(seq, t, lengths):
    if t >= lengths[seq]:
        return None
    g = torch.Generator().manual_seed(1000 * seq + t)
    n = 1 + (seq + t) % 3  # object count changes from frame to frame
    return torch.randn(n, 8, 3, generator=g), [f"o{j}" for j in range(n)], torch.randn(n, 4, 4, generator=g)


def _worker_tracking(rank, world, port, lengths, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = len(lengths)
        job = gdist.ShardedTracking(_StubTracker, S)
        assert list(job.owned()) == list(range(*gdist.shard_bounds(S, rank, world)))
        got = job.run(lambda s, t: _frames(s, t, lengths))
        # reference: every sequence alone, in one process
        ok = sorted(got) == list(range(S))
        for s in range(S):
            tr = _StubTracker(1)
            for t in range(lengths[s]):
                ref = tr.step([_frames(s, t, lengths)])[0]["average_sRT"]
                ok = ok and torch.equal(got[s][t]["average_sRT"], ref)
            ok = ok and len(got[s]) == lengths[s]
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lengths", [[3, 5, 2, 4, 4], [2], [3, 3]])
def test_tracking_dispatch_world2(lengths):
    """Whole sequences per rank, frames in order, ragged lengths, more ranks than sequences: every rank ends up with the results
    of ALL sequences and they equal the one-sequence-at-a-time runs."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_tracking, args=(2, port, lengths, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}
    assert [gdist.sequence_owner(s, 5, 2) for s in range(5)] == [0, 0, 0, 1, 1]
