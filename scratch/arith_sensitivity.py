"""How much hinges on the contraction convention of `dx*dx + dy*dy + dz*dz` (CPU, oracle only).

For synth.make_batch(N) (REAL275-shaped): per level, the fraction of clouds whose furthest-point-sampling indices / centre
COORDINATES / ball-query indices differ between conventions A, B, C; then, on the clouds that differ, the downstream effect on the
encoder's feature vector and on PC-20 poses (same weights, same noise).   python scratch/arith_sensitivity.py [N] > profiles/r5_arith_sensitivity.txt
"""
import sys
import numpy as np
import torch

from genpose_amd import synth
from oracle import genpose_oracle as go
from oracle import pn2_oracle as ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pts = synth.make_batch(N)
cfg = go.LIGHT_CFG
res = {}
for a in "ABC":
    cur = pts
    lv = []
    for l, npnt in enumerate([512, 256, 128]):
        idx, _ = ops.furthest_point_sampling(cur, npnt, arith=a)
        new = np.ascontiguousarray(np.take_along_axis(cur, idx[:, :, None].astype(np.int64), axis=1))
        bq = [ops.ball_query(cfg["radii"][l][s], cfg["nsamples"][l][s], cur, new, arith=a) for s in range(2)]
        lv.append((idx, new, bq, cur))
        cur = new
    res[a] = lv
print(f"# synth.make_batch({N}): clouds whose grouping differs between contraction conventions (oracle/pn2_ops.c, bit-exact == HIP kernels)")
print("# idx = selected indices differ; xyz = selected COORDINATES differ (an index change between duplicate points is invisible downstream);")
print("# bq|same = ball-query indices differ on clouds whose inputs (points and centres) are bit-identical under both conventions")
print(f"{'pair':5} {'level':5} {'fps idx':>9} {'fps xyz':>9} {'bq0|same':>9} {'bq1|same':>9}")
diff_clouds = {}
for x, y in (("A", "B"), ("C", "B"), ("A", "C")):
    any_xyz = np.zeros(N, bool)
    for l in range(3):
        ix, nx, bx, cx = res[x][l]
        iy, ny, by, cy = res[y][l]
        di = (ix != iy).any(axis=1)
        dx = (nx != ny).reshape(N, -1).any(axis=1)
        same_in = ~((cx != cy).reshape(N, -1).any(axis=1) | dx)
        db = [((bx[s] != by[s]).reshape(N, -1).any(axis=1) & same_in) for s in range(2)]
        any_xyz |= dx | db[0] | db[1]
        print(f"{x}/{y:3} {l:5d} {di.mean():9.4f} {dx.mean():9.4f} {db[0].sum():5d}/{same_in.sum():<4d} {db[1].sum():4d}/{same_in.sum():<4d}")
    diff_clouds[(x, y)] = np.nonzero(any_xyz)[0]
    print(f"{x}/{y}: {len(diff_clouds[(x, y)])} of {N} clouds ({len(diff_clouds[(x, y)]) / N:.2%}) see a different grouping somewhere")

# downstream, on the clouds that differ
sd = go.make_state_dict(0, "score")
K, steps = 50, 20
print("\n# downstream on the differing clouds (oracle, fp32): encoder feature vector [1024] and PC-20 poses (K = 50, same noise)")
print(f"{'pair':5} {'clouds':>6} {'feat max|d|/max|f|':>20} {'feat median':>12} {'rot max|d|':>11} {'rot p99':>9} {'trans max|d|/scale':>19}   (test tolerances: feat 2e-4, PC-20 1e-3)")
for (x, y), ids in diff_clouds.items():
    if len(ids) == 0:
        continue
    ids = ids[:64]
    p = torch.from_numpy(pts[ids])
    cen = p.mean(dim=1)
    gen = torch.Generator().manual_seed(1)
    R = len(ids) * K
    prior = torch.randn(R, 9, generator=gen)
    z1, z2 = torch.randn(steps, R, 9, generator=gen), torch.randn(steps, R, 9, generator=gen)
    out = {}
    for a in (x, y):
        with ops.use_arith(a):
            f = go.encoder_forward(sd, p)
            pose, _, _ = go.pred_func(sd, p, cen, K, "pc", prior, sampling_steps=steps, z_langevin=z1, z_predictor=z2)
        out[a] = (f.numpy(), pose.numpy())
    fd = np.abs(out[x][0] - out[y][0]).max(axis=1) / np.abs(out[y][0]).max()
    rd = np.abs(out[x][1][..., :6] - out[y][1][..., :6])
    td = np.abs(out[x][1][..., 6:] - out[y][1][..., 6:]).max() / np.abs(out[y][1][..., 6:]).max()
    print(f"{x}/{y:3} {len(ids):6d} {fd.max():20.3e} {np.median(fd):12.3e} {rd.max():11.3e} {np.quantile(rd, 0.99):9.2e} {td:19.3e}")
