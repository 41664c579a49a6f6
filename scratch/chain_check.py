"""Chain form of the score trunk (csrc/trunk_chain.h) against the tile form: agreement and launch time per plan.

    python scratch/chain_check.py [--quick]

Prints (1) max relative difference of gp_score_eval_plan(tile) to the 16-row tile form for score / energy, incl. a ragged tail and a
K small enough that a workgroup spans more clouds than are staged; (2) PC sampler end states per plan against the 32-row plan;
(3) microseconds per pc_step launch (captured graph, finish launch subtracted) per plan and row count.
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from genpose_amd import _lib  # noqa: E402
from genpose_amd.samplers import PCSampler  # noqa: E402
from genpose_amd.scorenet import ScoreNetHIP  # noqa: E402
from genpose_amd.sde import SIGMA_MAX, SIGMA_MIN  # noqa: E402
from genpose_amd.weights_synth import make_state_dict  # noqa: E402

PLANS = (16, 32, 64, 128)
FLOP_ROW = 0.5335e6
PEAK = 157.3


def eval_check(net):
    gen = torch.Generator().manual_seed(0)
    for B, K in ((257, 50), (1000, 3), (40, 50), (3, 50)):
        R = B * K
        cvec = torch.randn(B, 768, generator=gen).cuda()
        x = torch.randn(R, 9, generator=gen).cuda()
        t0 = torch.full((1,), 0.3, device="cuda")
        tvec = net.time_embed(t0)[0].contiguous()
        sigma = (SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** t0).contiguous()
        for mode in ("score", "energy"):
            ref = net.evaluate(cvec, K, x, tvec, sigma, mode, tile=16)
            torch.cuda.synchronize()
            line = []
            for tile in PLANS[1:]:
                try:
                    got = net.evaluate(cvec, K, x, tvec, sigma, mode, tile=tile)
                except _lib.GenposeHipError:
                    line.append(f"{tile}: n/a")  # the chain form needs the rows of a workgroup to span <= 4 clouds
                    continue
                torch.cuda.synchronize()
                err = float((got - ref).abs().max() / ref.abs().max())
                line.append(f"{tile}: {err:.1e}")
                assert err < 1e-5, (B, K, mode, tile, err)
            print(f"eval  B={B:5d} K={K:3d} {mode:6s} vs tile16  " + "  ".join(line), flush=True)


def pc_check(net):
    gen = torch.Generator().manual_seed(1)
    for G, B1, K, n in ((2, 128, 50, 20), (1, 100, 50, 12), (3, 64, 50, 8)):
        B = G * B1
        R = B * K
        cvec = torch.randn(B, 768, generator=gen).cuda()
        centre = torch.randn(B, 3, generator=gen).cuda()
        x0 = (torch.randn(R, 9, generator=gen) * 50).cuda()
        z1, z2 = torch.randn(n, R, 9, generator=gen).cuda(), torch.randn(n, R, 9, generator=gen).cuda()
        outs = {}
        for tile in PLANS:
            try:
                smp = PCSampler(net, B, K, n, "cuda", use_graph=False, record_traj=True, groups=G, tile=tile)
            except ValueError:
                continue
            xs, mean_x = smp.run(cvec, centre, x0, z1, z2)
            torch.cuda.synchronize()
            outs[tile] = (xs.clone(), mean_x.clone())
        ref = outs[32] if 32 in outs else outs[16]
        line = []
        for tile, (xs, mx) in outs.items():
            e1 = float((mx - ref[1]).abs().max() / ref[1].abs().max())
            e2 = float((xs - ref[0]).abs().max() / ref[0].abs().max())
            line.append(f"{tile}: {e1:.1e}/{e2:.1e}")
            assert torch.isfinite(mx).all() and e1 < 2e-3 and e2 < 2e-3, (G, B1, tile, e1, e2)
        print(f"pc    G={G} B1={B1} K={K} n={n}  mean_x/traj vs tile32  " + "  ".join(line), flush=True)


def time_plan(net, G, B1, K, n, tile):
    B = G * B1
    R = B * K
    try:
        smp = PCSampler(net, B, K, n, "cuda", use_graph=True, record_traj=False, groups=G, tile=tile)
    except ValueError:
        return None
    cvec = torch.randn(B, 768, device="cuda")
    centre = torch.randn(B, 3, device="cuda")
    x0 = torch.randn(R, 9, device="cuda") * 50
    smp.run(cvec, centre, x0)
    smp.run(cvec, centre, x0)
    torch.cuda.synchronize()
    reps = 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        smp.graph.replay()
    e1.record()
    torch.cuda.synchronize()
    chain = e0.elapsed_time(e1) * 1e-3 / reps
    gfin = torch.cuda.CUDAGraph()
    smp.launch_step(n)
    torch.cuda.synchronize()
    with torch.cuda.graph(gfin):
        for _ in range(50):
            smp.launch_step(n)
    gfin.replay()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record()
    gfin.replay()
    f1.record()
    torch.cuda.synchronize()
    fin = f0.elapsed_time(f1) * 1e-3 / 50
    per = (chain - fin) / n
    return per * 1e6, R * FLOP_ROW / per / 1e12 / PEAK


def timing(net, quick):
    shapes = [(10, 64, 50), (5, 64, 50), (1, 256, 50), (5, 256, 50), (2, 64, 50), (1, 64, 50), (20, 64, 50)]
    if quick:
        shapes = [shapes[0], shapes[5]]
    n = 40
    for G, B1, K in shapes:
        line = []
        for tile in PLANS:
            r = time_plan(net, G, B1, K, n, tile)
            line.append(f"{tile}: ---" if r is None else f"{tile}: {r[0]:6.1f} us ({r[1]:.3f})")
        auto = PCSampler(net, G * B1, K, 2, "cuda", use_graph=False, groups=G).tile
        print(f"time  G={G:2d} x {B1} clouds x {K} = {G * B1 * K:6d} rows  " + "  ".join(line) + f"   auto -> {auto}", flush=True)


def main():
    _lib.check_device()
    quick = "--quick" in sys.argv
    net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
    t = time.time()
    if "--time-only" not in sys.argv:
        eval_check(net)
        pc_check(net)
    timing(net, quick)
    print(f"chain_check done in {time.time() - t:.1f} s")


if __name__ == "__main__":
    main()
