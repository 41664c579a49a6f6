"""REAL275-shaped synthetic inputs: no dataset / checkpoint is reachable offline, so the bench,
the smoke test and the golden fixtures all use seeded synthetic clouds shaped like the reference's input:
single-view 2.5-D surfaces of bottle/bowl/laptop/box-like objects back-projected with the REAL camera
intrinsics (runners/evaluation_single.py:53-54), depth quantised to 1 mm (uint16 mm depth maps,
utils/sgpa_utils.py:194-211), pixel-deduplicated, then resampled to 1024 points exactly as
`sample_points` does (runners/evaluation_single.py:120-133: tile if fewer, random subset if more).
numpy only - importable without torch or the HIP library.
"""
import numpy as np

FX, FY, CX, CY = 591.0125, 590.16775, 322.525, 244.11084


def _rand_rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _surface(rng, family, n):
    """Points + outward normals on a unit-ish object surface."""
    if family == 0:  # cylinder (bottle / can / mug body)
        th = rng.uniform(0, 2 * np.pi, n)
        h = rng.uniform(-0.5, 0.5, n)
        r = 0.3
        p = np.stack([r * np.cos(th), h, r * np.sin(th)], 1)
        nrm = np.stack([np.cos(th), np.zeros(n), np.sin(th)], 1)
    elif family == 1:  # box (camera)
        face = rng.integers(0, 6, n)
        uv = rng.uniform(-0.5, 0.5, (n, 2))
        p = np.zeros((n, 3))
        nrm = np.zeros((n, 3))
        ext = np.array([0.5, 0.35, 0.3])
        for f in range(6):
            m = face == f
            ax, sgn = f // 2, 1 - 2 * (f % 2)
            o = [a for a in range(3) if a != ax]
            p[m, ax] = sgn * ext[ax]
            p[m, o[0]] = uv[m, 0] * 2 * ext[o[0]]
            p[m, o[1]] = uv[m, 1] * 2 * ext[o[1]]
            nrm[m, ax] = sgn
    elif family == 2:  # hemispherical bowl
        u = rng.uniform(0, 1, n)
        th = rng.uniform(0, 2 * np.pi, n)
        ph = np.arccos(u)  # upper hemisphere
        d = np.stack([np.sin(ph) * np.cos(th), -np.cos(ph), np.sin(ph) * np.sin(th)], 1)
        p = 0.5 * d
        nrm = d * np.where(rng.uniform(size=(n, 1)) < 0.5, 1.0, -1.0)  # inside and outside visible
    else:  # two-plane "laptop"
        half = rng.uniform(size=n) < 0.5
        uv = rng.uniform(-0.5, 0.5, (n, 2))
        p = np.zeros((n, 3))
        nrm = np.zeros((n, 3))
        p[half] = np.stack([uv[half, 0], np.zeros(half.sum()), uv[half, 1] * 0.7], 1)
        nrm[half] = [0, 1, 0]
        nh = ~half
        p[nh] = np.stack([uv[nh, 0], (uv[nh, 1] + 0.5) * 0.7, np.full(nh.sum(), -0.35)], 1)
        nrm[nh] = [0, 0, 1]
    return p, nrm


def sample_points(pcl, n_pts, rng):
    """Semantics of sample_points (evaluation_single.py:120-133)."""
    total = pcl.shape[0]
    if total < n_pts:
        pcl = np.concatenate([np.tile(pcl, (n_pts // total, 1)), pcl[: n_pts % total]], axis=0)
    elif total > n_pts:
        pcl = pcl[rng.permutation(total)[:n_pts]]
    return pcl


def make_cloud(i, n_pts=1024, force_few=None):
    """One REAL275-shaped cloud [n_pts,3] float32, seeded by its index."""
    rng = np.random.default_rng(1234 + i)
    family = int(rng.integers(0, 4))
    extent = rng.uniform(0.05, 0.30)
    centre = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.5, 1.2)])
    Rm = _rand_rot(rng)
    few = (rng.uniform() < 0.2) if force_few is None else force_few
    n_raw = int(rng.integers(100, 1023)) * 3 if few else 12000
    p, nrm = _surface(rng, family, n_raw)
    p = (p * extent) @ Rm.T + centre
    nrm = nrm @ Rm.T
    vis = np.sum(nrm * (-p), axis=1) > 0  # camera at the origin
    p = p[vis]
    if p.shape[0] < 8:
        p = (_surface(rng, 0, 500)[0] * extent) + centre
    u = np.round(p[:, 0] / p[:, 2] * FX + CX).astype(np.int64)
    v = np.round(p[:, 1] / p[:, 2] * FY + CY).astype(np.int64)
    z_mm = np.round(p[:, 2] * 1000.0).astype(np.int64)
    _, first = np.unique(u * 100000 + v, return_index=True)  # one depth sample per pixel
    first.sort()
    u, v, z_mm = u[first], v[first], z_mm[first]
    z = z_mm / 1000.0
    pcl = np.stack([(u - CX) * z / FX, (v - CY) * z / FY, z], 1)
    if few and pcl.shape[0] >= n_pts:
        pcl = pcl[: int(rng.integers(100, n_pts - 1))]
    return sample_points(pcl, n_pts, rng).astype(np.float32)


def make_batch(B, start=0, n_pts=1024):
    return np.stack([make_cloud(start + i, n_pts) for i in range(B)], 0)


def smoke_batch(B, seed=0, n_pts=1024):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(B, n_pts, 3)) * 0.05 + np.array([0, 0, 0.8])).astype(np.float32)


# ------------------------------------------------------------------ fixtures for tests/golden (deterministic)
def golden_clouds(seed=0):
    """[4,1024,3]: two surface clouds, one with ~300 unique points tiled to 1024 (exact duplicates -> d=0 FPS ties),
    one on an exact grid (equal-distance ties in FPS and at the ball-query radius)."""
    a = make_cloud(10_000 + seed, force_few=False)
    b = make_cloud(10_001 + seed, force_few=False)
    c = make_cloud(10_002 + seed, force_few=True)
    gx, gy = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    d = np.stack([gx.ravel() * 0.005 - 0.08, gy.ravel() * 0.005 - 0.08, np.full(1024, 0.75)], 1).astype(np.float32)
    return np.stack([a, b, c, d], 0).astype(np.float32)


def golden_tracking_frames(n_frames=3, n_obj=2):
    """[F,n_obj,1024,3]: the same objects drifting a few mm per frame."""
    base = np.stack([make_cloud(20_000 + o, force_few=False) for o in range(n_obj)], 0)
    out = []
    for f in range(n_frames):
        shift = np.array([0.004 * f, -0.002 * f, 0.003 * f], dtype=np.float32)
        out.append(base + shift)
    return np.stack(out, 0).astype(np.float32)


def golden_tracking_gt(n_obj=2):
    rng = np.random.default_rng(4242)
    RT = np.tile(np.eye(4, dtype=np.float32), (n_obj, 1, 1))
    frames0 = golden_tracking_frames(1, n_obj)[0]
    for o in range(n_obj):
        RT[o, :3, :3] = _rand_rot(rng)
        RT[o, :3, 3] = frames0[o].mean(0)
    return RT


def golden_map_results(seed=77, n_images=12, K=10):
    """Synthetic per-image detection + multi-hypothesis pose results in the reference's container layout (the dicts that
    evaluation_single.py:505-518 hands to compute_mAP): ground truth with NOCS-style scaled sRT, 2-D boxes, mug handle
    visibility; detections with missed / spurious / mis-classified instances; K pose hypotheses per detection scattered
    around the truth with energies that correlate with their quality."""
    rng = np.random.default_rng(seed)

    def rot_about(axis, deg):
        a = np.asarray(axis, dtype=np.float64)
        a = a / np.linalg.norm(a)
        t = np.deg2rad(deg)
        Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + np.sin(t) * Kx + (1 - np.cos(t)) * (Kx @ Kx)

    results = []
    for _ in range(n_images):
        n_gt = int(rng.integers(0, 6))
        gt_cls = rng.integers(1, 7, size=n_gt).astype(np.int32)
        gt_RTs = np.tile(np.eye(4), (n_gt, 1, 1))
        gt_boxes = np.zeros((n_gt, 4), dtype=np.int32)
        for g in range(n_gt):
            R = _rand_rot(rng).astype(np.float64)
            gt_RTs[g, :3, :3] = R * rng.uniform(0.1, 0.4)
            gt_RTs[g, :3, 3] = [rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.5, 1.2)]
            y1, x1 = int(rng.integers(0, 300)), int(rng.integers(0, 450))
            gt_boxes[g] = [y1, x1, y1 + int(rng.integers(40, 160)), x1 + int(rng.integers(40, 160))]
        hv = rng.integers(0, 2, size=n_gt).astype(np.int32)
        pc, pb, ps, hyp, en = [], [], [], [], []
        for g in range(n_gt):
            if rng.random() > 0.85:
                continue  # missed
            cls = int(gt_cls[g]) if rng.random() < 0.9 else int(rng.integers(1, 7))
            jit = rng.integers(-12, 13, size=4)
            pc.append(cls)
            pb.append(gt_boxes[g] + jit)
            ps.append(rng.uniform(0.3, 1.0))
            Rg = gt_RTs[g, :3, :3] / np.cbrt(np.linalg.det(gt_RTs[g, :3, :3]))
            h = np.tile(np.eye(4), (K, 1, 1))
            e = np.zeros((K, 2))
            spread = rng.choice([2.0, 6.0, 20.0])
            for k in range(K):
                ang = abs(rng.normal(0, spread)) if rng.random() < 0.85 else rng.uniform(30, 180)
                Rk = rot_about(rng.normal(size=3), ang) @ Rg
                if cls in (1, 2, 4):
                    Rk = Rk @ rot_about([0, 1, 0], rng.uniform(0, 360))  # free spin about the symmetry axis
                dt = rng.normal(0, 0.01 * spread / 2.0, size=3)
                h[k, :3, :3] = Rk
                h[k, :3, 3] = gt_RTs[g, :3, 3] + dt
                e[k] = [-ang + rng.normal(0, 3.0), -100 * np.linalg.norm(dt) + rng.normal(0, 1.0)]
            hyp.append(h)
            en.append(e)
        for _f in range(int(rng.integers(0, 3))):  # spurious detections
            pc.append(int(rng.integers(1, 7)))
            y1, x1 = int(rng.integers(0, 300)), int(rng.integers(0, 450))
            pb.append(np.array([y1, x1, y1 + 80, x1 + 80]))
            ps.append(rng.uniform(0.05, 0.6))
            h = np.tile(np.eye(4), (K, 1, 1))
            for k in range(K):
                h[k, :3, :3] = _rand_rot(rng)
                h[k, :3, 3] = [rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.5, 1.2)]
            hyp.append(h)
            en.append(rng.normal(size=(K, 2)))
        n_p = len(pc)
        results.append({
            "gt_class_ids": gt_cls, "gt_bboxes": gt_boxes, "gt_RTs": gt_RTs, "gt_scales": np.ones((n_gt, 3)), "gt_handle_visibility": hv,
            "pred_class_ids": np.array(pc, dtype=np.int32), "pred_bboxes": np.array(pb, dtype=np.int32).reshape(n_p, 4),
            "pred_scores": np.array(ps, dtype=np.float64), "pred_RTs": np.tile(np.eye(4), (n_p, 1, 1)), "pred_scales": np.ones((n_p, 3)),
            "multi_hypothesis_pred_RTs": np.array(hyp).reshape(n_p, K, 4, 4), "energy": np.array(en).reshape(n_p, K, 2)})
    return results


def golden_depth_frame(seed=5, n_inst=6, H=480, W=640):
    """A synthetic REAL275-like frame for the depth -> cloud pre-processing: uint16 depth in mm (a tilted background plane
    with holes, elliptical objects in front of it), Mask-RCNN style outputs masks [H,W,n] bool, rois [n,4] (y1,x1,y2,x2),
    class_ids [n] (1..6).  Instances: large (> 1024 pixels), small (< 1024), touching the image border, an empty mask and a
    mask that lies on a depth hole (both skipped by the reference)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (900 + 0.4 * xx + 0.7 * yy).astype(np.float64)
    holes = rng.random((H, W)) < 0.03
    masks = np.zeros((H, W, n_inst), dtype=bool)
    rois = np.zeros((n_inst, 4), dtype=np.int32)
    specs = [(240, 320, 70, 90), (100, 520, 2, 3), (30, 40, 45, 60), (455, 610, 40, 50), (300, 100, 0, 0), (200, 200, 20, 20)]
    for i in range(n_inst):
        cy, cx, ry, rx = specs[i % len(specs)]
        if ry == 0:
            rois[i] = [cy - 5, cx - 5, cy + 5, cx + 5]  # detector box without mask pixels
            continue
        m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        masks[:, :, i] = m
        obj = 600 + 40 * i + 30 * np.sqrt(np.clip(1 - ((yy - cy) / ry) ** 2 - ((xx - cx) / rx) ** 2, 0, 1))
        if i % len(specs) == 5:
            depth[m] = 0  # sensor hole over the whole object
        else:
            depth[m] = obj[m]
        ys, xs = np.where(m)
        rois[i] = [max(ys.min() - 3, 0), max(xs.min() - 3, 0), min(ys.max() + 3, H - 1), min(xs.max() + 3, W - 1)]
    depth[holes] = 0
    class_ids = (np.arange(n_inst) % 6 + 1).astype(np.int32)
    return depth.astype(np.uint16), masks, rois, class_ids
