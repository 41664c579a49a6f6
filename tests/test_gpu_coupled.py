"""GPU: "faithful" multi-GPU coupling of the samplers (SURVEY §8e caveat).  A batch sharded over two ranks - here two processes
sharing the one device, collectives on gloo - with the per-step all-reduce of the gradient-norm sums (PC) / the per-attempt
all-reduce of the error-norm sums (RK45) gives every shard what the UNSHARDED batch gives it; without the coupling the shards visibly
step differently (their own batch statistics).  On RCCL the reductions are captured inside the samplers' hipGraphs: exercised with a
one-rank 'nccl' group (the box has one GPU)."""
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
        # PF-ODE: the RK45 error norm over the whole batch
        from genpose_amd.samplers import ODESampler
        y0 = x0[rows].cuda() * 0.04
        for tag, group in (("ode_coupled", dist.group.WORLD), ("ode_alone", None)):
            ode = ODESampler(net, bs, K, "cuda", coupling_group=group)
            _, xo = ode.run(args[0], args[1], y0, 0.4)
            torch.cuda.synchronize()
            np.save(os.path.join(out_dir, f"{tag}_{rank}.npy"), xo.cpu().numpy())
            np.save(os.path.join(out_dir, f"{tag}_sched_{rank}.npy"), np.stack([ode.last_stats["log_h"], ode.last_stats["log_err"]]))
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


def test_sharded_ode_batch_with_coupled_error_norm_equals_the_unsharded_batch(tmp_path):
    """cond_ode_sampler's solve_ivp takes ONE step size for the whole batch (RMS error norm over all R*9 components).  Two shards
    whose controllers decide on the all-reduced sums take the unsharded batch's accept / reject sequence, attempt for attempt."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from genpose_amd.samplers import ODESampler
    from genpose_amd.scorenet import ScoreNetHIP
    from genpose_amd.weights_synth import make_state_dict
    net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
    cvec, centre, x0, _, _ = _inputs()
    ode = ODESampler(net, B, K, "cuda")
    _, full = ode.run(cvec.cuda(), centre.cuda(), x0.cuda() * 0.04, 0.4)
    torch.cuda.synchronize()
    full = full.cpu().numpy()
    sched_full = np.stack([ode.last_stats["log_h"], ode.last_stats["log_err"]])
    coupled = np.concatenate([np.load(tmp_path / f"ode_coupled_{r}.npy") for r in range(2)])
    alone = np.concatenate([np.load(tmp_path / f"ode_alone_{r}.npy") for r in range(2)])
    for r in range(2):
        sc = np.load(tmp_path / f"ode_coupled_sched_{r}.npy")
        assert sc.shape == sched_full.shape, (sc.shape, sched_full.shape)  # the same number of attempts on every shard
        np.testing.assert_allclose(sc[0], sched_full[0], rtol=1e-6)        # the same step sizes
        np.testing.assert_allclose(sc[1], sched_full[1], rtol=1e-4, atol=1e-9)  # the same error norms (sums formed in a different order)
    s0, s1 = np.load(tmp_path / "ode_alone_sched_0.npy"), np.load(tmp_path / "ode_alone_sched_1.npy")
    assert s0.shape != s1.shape or not np.allclose(s0[0], s1[0], rtol=1e-3)  # uncoupled shards run their own step control
    scale = np.abs(full[:, 6:]).max()
    np.testing.assert_allclose(coupled[:, :6], full[:, :6], rtol=0, atol=1e-5)
    np.testing.assert_allclose(coupled[:, 6:], full[:, 6:], rtol=0, atol=1e-6 * max(1.0, scale))
    assert np.abs(alone - full).max() > 10 * np.abs(coupled - full).max()


def _nccl_one_rank(rank, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from genpose_amd.samplers import ODESampler, PCSampler
        from genpose_amd.scorenet import ScoreNetHIP
        from genpose_amd.weights_synth import make_state_dict
        net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
        cvec, centre, x0, z1, z2 = (t.cuda() for t in _inputs())
        res = {}
        for tag, group in (("coupled", dist.group.WORLD), ("alone", None)):
            smp = PCSampler(net, B, K, N, "cuda", coupling_group=group)
            for _ in range(2):
                _, m = smp.run(cvec, centre, x0, z1, z2)
            torch.cuda.synchronize()
            res[f"pc_{tag}"] = m.cpu().numpy()
            res[f"pc_{tag}_graph"] = np.array(smp.graph is not None)
            ode = ODESampler(net, B, K, "cuda", coupling_group=group)
            for _ in range(2):
                _, xo = ode.run(cvec, centre, x0 * 0.04, 0.4)
            torch.cuda.synchronize()
            res[f"ode_{tag}"] = xo.cpu().numpy()
            res[f"ode_{tag}_graph"] = np.array(bool(ode._graphs.get("graph")))
            res[f"ode_{tag}_attempts"] = np.array(int(ode.last_stats["n_attempts"]))
        np.savez(os.path.join(out_dir, "nccl1.npz"), **res)
    finally:
        dist.destroy_process_group()


def test_coupling_reductions_are_captured_in_the_graphs_on_rccl(tmp_path):
    """Backend 'nccl' = RCCL: the coupled samplers keep their hipGraphs - the per-step / per-attempt all-reduce is captured with the
    launches.  One rank (the box has one GPU): the reduction is the identity, so the coupled result must equal the uncoupled one."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_nccl_one_rank, args=(port, str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl1.npz")
    assert bool(r["pc_coupled_graph"]) and bool(r["ode_coupled_graph"]) and bool(r["pc_alone_graph"])
    np.testing.assert_allclose(r["pc_coupled"], r["pc_alone"], rtol=0, atol=1e-5 * np.abs(r["pc_alone"]).max())
    assert int(r["ode_coupled_attempts"]) == int(r["ode_alone_attempts"])
    np.testing.assert_allclose(r["ode_coupled"], r["ode_alone"], rtol=0, atol=1e-9 * max(1.0, np.abs(r["ode_alone"]).max()))
