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
