"""GPU: "faithful" multi-GPU coupling of the PC sampler (SURVEY §8e caveat).  A batch sharded over two ranks - here two processes
sharing the one device, collectives on gloo - with the per-step all-reduce of the gradient-norm sums gives every shard what the
UNSHARDED batch gives it; without the coupling the shards visibly step differently (their own batch means)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

B, K, N = 8, 10, 12


def _inputs():
    gen = torch.Generator().manual_seed(31)
    cvec = torch.randn(B, 768, generator=gen)
    centre = torch.randn(B, 3, generator=gen) * 0.3
    x0 = torch.randn(B * K, 9, generator=gen) * 50.0
    x0[B * K // 2:] *= 0.1  # the two shards see very different gradient norms
    z1, z2 = torch.randn(N, B * K, 9, generator=gen), torch.randn(N, B * K, 9, generator=gen)
    return cvec, centre, x0, z1, z2


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from genpose_amd.samplers import PCSampler
        from genpose_amd.scorenet import ScoreNetHIP
        from genpose_amd.weights_synth import make_state_dict
        net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
        cvec, centre, x0, z1, z2 = _inputs()
        bs = B // world
        cl, rows = slice(rank * bs, (rank + 1) * bs), slice(rank * bs * K, (rank + 1) * bs * K)
        args = (cvec[cl].cuda(), centre[cl].cuda(), x0[rows].cuda(), z1[:, rows].contiguous().cuda(), z2[:, rows].contiguous().cuda())
        for tag, group in (("coupled", dist.group.WORLD), ("alone", None)):
            smp = PCSampler(net, bs, K, N, "cuda", coupling_group=group)
            _, m = smp.run(*args)
            torch.cuda.synchronize()
            np.save(os.path.join(out_dir, f"{tag}_{rank}.npy"), m.cpu().numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_batch_with_coupling_equals_the_unsharded_batch(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from genpose_amd.samplers import PCSampler
    from genpose_amd.scorenet import ScoreNetHIP
    from genpose_amd.weights_synth import make_state_dict
    net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
    cvec, centre, x0, z1, z2 = _inputs()
    _, full = PCSampler(net, B, K, N, "cuda").run(cvec.cuda(), centre.cuda(), x0.cuda(), z1.cuda(), z2.cuda())
    torch.cuda.synchronize()
    full = full.cpu().numpy()
    coupled = np.concatenate([np.load(tmp_path / f"coupled_{r}.npy") for r in range(2)])
    alone = np.concatenate([np.load(tmp_path / f"alone_{r}.npy") for r in range(2)])
    scale = np.abs(full).max()
    # the all-reduced sum is formed in a different order than the single-launch reduction: fp32 round-off over 12 steps
    np.testing.assert_allclose(coupled, full, rtol=0, atol=1e-4 * scale)
    assert np.abs(alone - full).max() > 1e-2 * scale  # shard-local means are a different sampler
