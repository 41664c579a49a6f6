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
    return _through_the_camera(p, rng, few, n_pts)


def _through_the_camera(p, rng, few, n_pts):
    """Visible surface points [m,3] (camera frame, metres) -> what the reference's loader would hand over: projected with the REAL
    intrinsics, snapped to pixels, depth quantised to 1 mm, one sample per pixel, back-projected, resampled to n_pts."""
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


# ------------------------------------------------------------------ clouds WITH a ground-truth pose (training / accuracy proxy)
# NOCS categories (class id = index + 1, utils/sgpa_utils.py synset_names); bottle / bowl / can are symmetric about the object's y axis,
# which is how the reference scores them (utils/metrics.py:107-113, sgpa_utils.py:548-560); camera / laptop / mug are built WITHOUT a
# rotational symmetry (off-centre lens, lid shorter than the base, handle) so that their full rotation is identifiable from one view.
CATEGORIES = ("bottle", "bowl", "camera", "can", "laptop", "mug")


def _ring(rng, n, r0, r1, y0, y1, two_sided=False):
    """Surface of revolution about y between (r0, y0) and (r1, y1) (cylinder, cone or disc), sampled uniformly in area."""
    u = rng.uniform(size=n)
    if r0 != r1:
        rr = np.sqrt(u * (r1 * r1 - r0 * r0) + r0 * r0)  # area-uniform along a cone / annulus
        s = (rr - r0) / (r1 - r0)
    else:
        rr, s = np.full(n, r0), u
    th = rng.uniform(0, 2 * np.pi, n)
    p = np.stack([rr * np.cos(th), y0 + s * (y1 - y0), rr * np.sin(th)], 1)
    dr, dy = r1 - r0, y1 - y0
    L = np.hypot(dr, dy)
    nr, ny = dy / L, -dr / L  # outward for dy > 0; a disc (dy = 0) gets +-y
    nrm = np.stack([nr * np.cos(th), np.full(n, ny), nr * np.sin(th)], 1)
    if two_sided:
        nrm = nrm * np.where(rng.uniform(size=(n, 1)) < 0.5, 1.0, -1.0)
    return p, nrm


def _pieces(rng, n, pieces):
    """pieces: [(area weight, sampler(rng, m))] -> n points over all of them in proportion to their areas."""
    w = np.array([a for a, _ in pieces], dtype=np.float64)
    counts = rng.multinomial(n, w / w.sum())
    out = [f(rng, int(m)) for (_, f), m in zip(pieces, counts) if m > 0]
    return np.concatenate([o[0] for o in out], 0), np.concatenate([o[1] for o in out], 0)


def _rect(rng, n, origin, eu, ev, normal, two_sided=False):
    uv = rng.uniform(0, 1, (n, 2))
    p = np.asarray(origin, dtype=np.float64) + uv[:, :1] * np.asarray(eu, dtype=np.float64) + uv[:, 1:] * np.asarray(ev, dtype=np.float64)
    nrm = np.tile(np.asarray(normal, dtype=np.float64), (n, 1))
    if two_sided:
        nrm = nrm * np.where(rng.uniform(size=(n, 1)) < 0.5, 1.0, -1.0)
    return p, nrm


def _posed_surface(rng, cat, n):
    """Object-frame points + normals of category `cat` (index into CATEGORIES); y is the object's up axis as in NOCS."""
    cone = lambda r0, r1, y0, y1: np.pi * (r0 + r1) * np.hypot(r1 - r0, y1 - y0)
    if cat == 0:    # bottle: body, shoulder, neck, cap, bottom - up and down differ
        return _pieces(rng, n, [(cone(.25, .25, -.5, .15), lambda g, m: _ring(g, m, .25, .25, -.5, .15)),
                                (cone(.25, .10, .15, .30), lambda g, m: _ring(g, m, .25, .10, .15, .30)),
                                (cone(.10, .10, .30, .50), lambda g, m: _ring(g, m, .10, .10, .30, .50)),
                                (cone(.10, 1e-3, .50, .50), lambda g, m: _ring(g, m, .10, 1e-3, .50, .50)),
                                (cone(1e-3, .25, -.5, -.5), lambda g, m: _ring(g, m, 1e-3, .25, -.5, -.5))])
    if cat == 1:    # bowl: hemisphere open towards +y, inside and outside visible
        u = rng.uniform(0, 1, n)
        th = rng.uniform(0, 2 * np.pi, n)
        ph = np.arccos(u)
        d = np.stack([np.sin(ph) * np.cos(th), -np.cos(ph), np.sin(ph) * np.sin(th)], 1)
        return 0.5 * d, d * np.where(rng.uniform(size=(n, 1)) < 0.5, 1.0, -1.0)
    if cat == 2:    # camera: box with an off-centre lens barrel on the +z face
        ex, ey, ez = 0.5, 0.35, 0.3
        faces = []
        for ax, (a, b, c) in enumerate(((ex, ey, ez), (ey, ez, ex), (ez, ex, ey))):
            for sgn in (1.0, -1.0):
                o, eu, ev, nr = np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3)
                i, j, k = ax, (ax + 1) % 3, (ax + 2) % 3
                o[i], o[j], o[k] = sgn * a, -b, -c
                eu[j], ev[k], nr[i] = 2 * b, 2 * c, sgn
                faces.append((4 * b * c, (lambda o, eu, ev, nr: lambda g, m: _rect(g, m, o, eu, ev, nr))(o, eu, ev, nr)))

        def lens(g, m, kind):
            q, nq = _ring(g, m, .2, .2, ez, ez + .3) if kind == 0 else _ring(g, m, 1e-3, .2, ez + .3, ez + .3)
            if kind == 1:
                nq = -nq  # the cap of the barrel looks outwards (+ along the barrel)
            # the barrel's axis is the object's z: (x, y, z) <- (x, z, y) of the ring, then the offset in the face
            q, nq = q[:, [0, 2, 1]], nq[:, [0, 2, 1]]
            q[:, 0] += 0.2
            q[:, 1] -= 0.08
            return q, nq
        return _pieces(rng, n, faces + [(cone(.2, .2, 0, .3), lambda g, m: lens(g, m, 0)), (np.pi * .04, lambda g, m: lens(g, m, 1))])
    if cat == 3:    # can: slightly tapered (top narrower), lids
        return _pieces(rng, n, [(cone(.30, .24, -.5, .5), lambda g, m: _ring(g, m, .30, .24, -.5, .5)),
                                (np.pi * .24 ** 2, lambda g, m: (lambda q: (q[0], -q[1]))(_ring(g, m, 1e-3, .24, .5, .5))),
                                (np.pi * .30 ** 2, lambda g, m: _ring(g, m, 1e-3, .30, -.5, -.5))])
    if cat == 4:    # laptop: base (1.0 x 0.7) and a shorter lid (1.0 x 0.5) hinged at the back, both thin (seen from either side)
        return _pieces(rng, n, [(0.70, lambda g, m: _rect(g, m, [-.5, 0, -.35], [1, 0, 0], [0, 0, .7], [0, 1, 0], True)),
                                (0.50, lambda g, m: _rect(g, m, [-.5, 0, -.35], [1, 0, 0], [0, .5, 0], [0, 0, 1], True))])
    if cat == 5:    # mug: open cylinder (inside visible), bottom, handle (half a torus in the xy plane on the +x side)
        def handle(g, m):
            a = g.uniform(-np.pi / 2, np.pi / 2, m)   # along the handle
            b = g.uniform(0, 2 * np.pi, m)            # around the tube
            R0, r0 = 0.22, 0.05
            c = np.stack([0.3 + R0 * np.cos(a), R0 * np.sin(a), np.zeros(m)], 1)
            radial = np.stack([np.cos(a), np.sin(a), np.zeros(m)], 1)
            nq = radial * np.cos(b)[:, None] + np.array([0, 0, 1.0]) * np.sin(b)[:, None]
            return c + r0 * nq, nq
        return _pieces(rng, n, [(cone(.3, .3, -.4, .4), lambda g, m: _ring(g, m, .3, .3, -.4, .4, True)),
                                (np.pi * .09, lambda g, m: _ring(g, m, 1e-3, .3, -.4, -.4, True)),
                                (np.pi * .22 * 2 * np.pi * .05, handle)])
    raise ValueError(cat)


def make_posed_cloud(i, n_pts=1024, cat=None):
    """One REAL275-shaped cloud with its ground truth, seeded by its index: {'pts' [n_pts,3] f32 (camera frame, metres), 'R' [3,3],
    't' [3] (object frame -> camera frame; the 9-D pose of the reference's 'rot_matrix' mode is [R[:,0], R[:,1], t],
    datasets_genpose.py:692-695), 'cat' 0..5, 'scale'}.  Same camera model and resampling as make_cloud; 10 % of the instances have
    fewer than 1024 distinct points."""
    rng = np.random.default_rng([0x706F7365, i])
    cat = int(rng.integers(0, len(CATEGORIES))) if cat is None else int(cat)
    extent = rng.uniform(0.08, 0.30)
    centre = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.5, 1.2)])
    Rm = _rand_rot(rng)
    few = bool(rng.uniform() < 0.1)
    p, nrm = _posed_surface(rng, cat, int(rng.integers(100, 1023)) * 3 if few else 12000)
    p = (p * extent) @ Rm.T + centre
    vis = np.sum((nrm @ Rm.T) * (-p), axis=1) > 0  # camera at the origin
    if vis.sum() >= 8:
        p = p[vis]
    return {"pts": _through_the_camera(p, rng, few, n_pts), "R": Rm, "t": centre, "cat": cat, "scale": extent}


def posed_batch(indices, n_pts=1024):
    """-> dict of arrays with leading dim len(indices): pts f32, gt_pose [.,9] f32, R, t f64, cat i64, scale, handle_visibility (1)."""
    items = [make_posed_cloud(int(i), n_pts) for i in indices]
    R = np.stack([it["R"] for it in items], 0)
    t = np.stack([it["t"] for it in items], 0)
    return {"pts": np.stack([it["pts"] for it in items], 0), "R": R, "t": t,
            "gt_pose": np.concatenate([R[:, :, 0], R[:, :, 1], t], axis=1).astype(np.float32),
            "cat": np.array([it["cat"] for it in items], dtype=np.int64), "scale": np.array([it["scale"] for it in items]),
            "handle_visibility": np.ones(len(items), dtype=np.int64)}


def posed_sequence(seed, n_frames=6, n_obj=5, n_pts=1024):
    """A short synthetic tracking sequence with ground truth: n_obj objects (categories drawn once), each moving smoothly - a few millimetres and
    about a degree per frame - and re-rendered through the camera model every frame (new visible side, new pixel grid, new resampling).
    -> dict: pts [F,n_obj,n_pts,3] f32, R [F,n_obj,3,3], t [F,n_obj,3] f64, cat [n_obj] i64."""
    rng = np.random.default_rng([0x73657175, seed])
    cats = rng.integers(0, len(CATEGORIES), n_obj)
    ext = rng.uniform(0.10, 0.28, n_obj)
    t0 = np.stack([rng.uniform(-0.25, 0.25, n_obj), rng.uniform(-0.25, 0.25, n_obj), rng.uniform(0.6, 1.1, n_obj)], 1)
    R0 = np.stack([_rand_rot(rng) for _ in range(n_obj)], 0)
    vel = rng.normal(size=(n_obj, 3)) * 0.003                      # metres per frame
    axis = rng.normal(size=(n_obj, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    rate = np.deg2rad(rng.uniform(0.5, 1.5, n_obj))                # radians per frame
    out = {"pts": np.zeros((n_frames, n_obj, n_pts, 3), np.float32), "R": np.zeros((n_frames, n_obj, 3, 3)), "t": np.zeros((n_frames, n_obj, 3)),
           "cat": cats.astype(np.int64)}
    for f in range(n_frames):
        for o in range(n_obj):
            a, th = axis[o], rate[o] * f
            Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            Rf = (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)) @ R0[o]
            tf = t0[o] + vel[o] * f
            p, nrm = _posed_surface(rng, int(cats[o]), 12000)
            p = (p * ext[o]) @ Rf.T + tf
            vis = np.sum((nrm @ Rf.T) * (-p), axis=1) > 0
            if vis.sum() >= 8:
                p = p[vis]
            out["pts"][f, o] = _through_the_camera(p, rng, False, n_pts)
            out["R"][f, o], out["t"][f, o] = Rf, tf
    return out


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
