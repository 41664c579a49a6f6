"""GPU: BASELINE configs[3] rehearsed at its real size - 2048 clouds sharded over EIGHT ranks (`dist.ShardedInference`: contiguous balanced
shards of 256 clouds, every kernel rank-local, one all-gather per result tensor).  The box has one GPU: the eight processes share cuda:0
and the collectives run on gloo; the sharding, the gather and the coupling logic are what runs on an 8-GPU node over RCCL.

 * plain sharding (the default): a shard IS one of the reference's own 256-cloud batches (evaluation_single.py:380-382 slices the
   instances into batches of `batch_size` = 256, each with its own batch-global solver statistics), so the gathered result must equal the
   one-process run over all 2048 clouds BIT FOR BIT - the drop-in call sequence pred_func (ODE, T0 = 0.55, K = 50) -> get_energy ->
   rank_aggregate of scripts/eval_single.sh;
 * faithful coupling (`net.coupling_group`): the 2048 clouds as ONE batch of the PC-100 sampler spread over the eight ranks, the Langevin
   step size from the all-reduced gradient-norm sums - every shard must get what the unsharded 102 400-row batch gives it, to fp32
   round-off (PC-100 tolerance of tests/test_gpu_fullsize.py).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

WORLD, NCL, K, T0, NSTEPS = 8, 2048, 50, 0.55, 100
SHARD = NCL // WORLD


def _agents(sampler, steps):
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    from genpose_amd.weights_synth import make_state_dict
    sa = PoseNet(get_config(posenet_mode="score", sampler_mode=[sampler], sampling_steps=steps))
    sa.load_state_dict(make_state_dict(0, "score"))
    ea = PoseNet(get_config(posenet_mode="energy"))
    ea.load_state_dict(make_state_dict(0, "energy"))
    return sa, ea


def _shard_prior(j):
    """Standard-normal prior draws of shard / batch j (the reference draws them on the CPU generator, sde.py:28)."""
    return torch.randn(SHARD * K, 9, generator=torch.Generator().manual_seed(1000 + j))


def _shard_noise(j):
    g = torch.Generator(device="cuda").manual_seed(2000 + j)
    return (torch.randn(NSTEPS, SHARD * K, 9, generator=g, device="cuda"), torch.randn(NSTEPS, SHARD * K, 9, generator=g, device="cuda"))


def _drop_in(sa, ea, shard_ids):
    """The eval_single call sequence over consecutive 256-cloud batches `shard_ids` of one process."""
    from genpose_amd.runner import SingleFrameRunner
    order = iter(shard_ids)
    sa.net.prior_fn = lambda shape, T=1.0: _shard_prior(next(order)) * (0.01 * 5000.0 ** T)
    runner = SingleFrameRunner(sa, ea, repeat_num=K, T0=T0, batch_size=SHARD)
    return lambda clouds: {k: v for k, v in runner.infer_tensors(clouds).items() if k in ("pred_pose", "energy", "average_sRT")}


def _coupled_pc(sa, ea, shard_ids, group):
    from genpose_amd import reward
    sa.net.coupling_group = group
    ids = list(shard_ids)
    sa.net.prior_fn = lambda shape, T=1.0: torch.cat([_shard_prior(j) for j in ids]) * (0.01 * 5000.0 ** T)

    def infer(clouds):
        data = {"pts": clouds, "pts_center": clouds.mean(dim=1)}
        nz = [_shard_noise(j) for j in ids]
        noise = (torch.cat([z[0] for z in nz], dim=1), torch.cat([z[1] for z in nz], dim=1))
        pred = sa.pred_func(data, repeat_num=K, save_path=None, noise=noise)
        energy = ea.get_energy(data=data, pose_samples=pred, T=1e-5)
        return {"pred_pose": pred, "energy": energy, "avg_pose": reward.rank_aggregate(pred, energy, ratio=0.6)["avg_pose"]}
    return infer


def _worker(rank, port, out_dir):
    import time
    torch.set_num_threads(4)  # eight processes share the host: the default (one thread per core, each) made this fixture 160 s of thrashing
    marks = [("start", time.time())]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    marks.append(("process group", time.time()))
    try:
        torch.cuda.set_device(0)
        from genpose_amd.dist import ShardedInference
        clouds = torch.from_numpy(np.load(os.path.join(out_dir, "clouds.npy"))).cuda()
        sa, ea = _agents("ode", None)
        torch.cuda.synchronize()
        marks.append(("device context + ODE agents", time.time()))
        out = ShardedInference(_drop_in(sa, ea, [rank]))(clouds)
        torch.cuda.synchronize()
        marks.append(("sharded drop-in sequence", time.time()))
        sp, ep = _agents("pc", NSTEPS)
        outc = ShardedInference(_coupled_pc(sp, ep, [rank], dist.group.WORLD))(clouds)
        torch.cuda.synchronize()
        marks.append(("coupled PC-100", time.time()))
        if rank == 0:
            with open(os.path.join(out_dir, "worker0_times.txt"), "w") as f:
                f.write(", ".join(f"{n} {t - marks[i][1]:.1f} s" for i, (n, t) in enumerate(marks[1:])))
        if rank == WORLD - 1:  # every rank holds the full result set: take it from the LAST rank
            np.savez(os.path.join(out_dir, "sharded.npz"), **{k: v.cpu().numpy() for k, v in out.items()},
                     **{"c_" + k: v.cpu().numpy() for k, v in outc.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def sharded(tmp_path_factory):
    from genpose_amd import synth
    d = str(tmp_path_factory.mktemp("dist8"))
    np.save(os.path.join(d, "clouds.npy"), synth.make_batch(NCL, start=9000))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import time
    t0 = time.time()
    mp.spawn(_worker, args=(port, d), nprocs=WORLD, join=True)
    line = f"[dist8 fixture] {WORLD} workers on one device: {time.time() - t0:.1f} s; rank 0: {open(os.path.join(d, 'worker0_times.txt')).read()}"
    print(line)
    scratch = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(scratch):
        with open(os.path.join(scratch, "dist8_fixture_times.txt"), "w") as f:
            f.write(line + "\n")
    return torch.from_numpy(np.load(os.path.join(d, "clouds.npy"))).cuda(), dict(np.load(os.path.join(d, "sharded.npz")))


def test_2048_clouds_over_eight_ranks_equal_the_one_process_run(sharded):
    clouds, got = sharded
    sa, ea = _agents("ode", None)
    want = _drop_in(sa, ea, range(WORLD))(clouds)
    torch.cuda.synchronize()
    for k, v in want.items():
        assert got[k].shape == tuple(v.shape) and got[k].shape[0] == NCL, k
        assert np.array_equal(got[k], v.cpu().numpy()), k  # a shard is one of the reference's own 256-cloud batches: bit for bit


def test_2048_cloud_batch_coupled_over_eight_ranks_equals_the_unsharded_batch(sharded):
    clouds, got = sharded
    sp, ep = _agents("pc", NSTEPS)
    want = {k: v.cpu().numpy() for k, v in _coupled_pc(sp, ep, range(WORLD), None)(clouds).items()}
    assert sp.net.last_sampler.R == NCL * K and sp.net.last_sampler.groups == 1  # ONE 102 400-row batch with one batch-global statistic
    pred, ref = got["c_pred_pose"], want["pred_pose"]
    assert pred.shape == ref.shape == (NCL, K, 9) and np.isfinite(pred).all()
    rot = np.abs(pred[..., :6] - ref[..., :6])
    # PC-100 renormalises the rotation columns every step, which amplifies round-off for the rare row whose column passes near zero
    # (tests/test_gpu_fullsize.py: p99.9 = 2.6e-5, max 7e-3 between two launch plans of the same batch)
    assert np.quantile(rot, 0.999) < 1e-3 and rot.max() < 2e-2, (np.quantile(rot, 0.999), rot.max())
    np.testing.assert_allclose(pred[..., 6:], ref[..., 6:], rtol=0, atol=1e-4 * np.abs(ref[..., 6:]).max())
    de = np.abs(got["c_energy"] - want["energy"]) / np.abs(want["energy"]).max()  # energies follow the poses: the same rare rows stand out
    assert np.quantile(de, 0.999) < 1e-3, np.quantile(de, 0.999)
    assert got["c_avg_pose"].shape == (NCL, 7) and np.isfinite(got["c_avg_pose"]).all()
    # shard-local statistics are a different sampler: a shard on its own must NOT reproduce the coupled result
    sp2, ep2 = _agents("pc", NSTEPS)
    alone = _coupled_pc(sp2, ep2, [3], None)(clouds[3 * SHARD:4 * SHARD])["pred_pose"].cpu().numpy()
    assert np.abs(alone[..., 6:] - ref[3 * SHARD:4 * SHARD, :, 6:]).max() > 10 * np.abs(pred[..., 6:] - ref[..., 6:]).max()
