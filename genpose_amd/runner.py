"""Runner loops over the agent: the batch / frame loops of runners/evaluation_single.py:309-489 and
runners/evaluation_tracking.py:262-337 restated around the HIP agent.

Inputs  : lists / arrays of [1024,3] float32 clouds (camera frame, metres, NOT centred); for tracking also per-frame
          `model_name`s and an initial sRT per object.
Outputs : multi_hypothesis_pred_RTs [n,K,4,4] (float64, as the reference's numpy RTs), energy [n,K,2], aggregated
          sRT [n,4,4].
Detection pre-processing (depth+mask -> cloud) and mAP evaluation are the "next" rows of SURVEY §8f.
"""
import math

import numpy as np
import torch

from . import reward, rotation
from .lru import ShapeCache


def make_batch_sample(pts):
    """The dict the agents consume (evaluation_single.py:394-403): `pts` stays un-centred, `pts_center` is the mean."""
    pts = pts.float()
    centre = torch.mean(pts[:, :, :3], dim=1)
    return {"pts": pts, "zero_mean_pts": pts - centre.unsqueeze(1), "pts_center": centre}


class _one_cpu_thread:
    """Host-side torch ops inside the per-frame loops run single-threaded: a multi-threaded CPU op leaves its OpenMP team spinning
    next to the HIP runtime's progress thread and the following graph replays stall for tens of milliseconds (EXPERIMENTS.md §G, the tracking history of DESIGN r5 §8)."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(1)

    def __exit__(self, *exc):
        torch.set_num_threads(self.n)


class SingleFrameRunner:
    """inference_pose + inference_energy (evaluation_single.py:356-489) without the pickle round trip in between."""

    def __init__(self, score_agent, energy_agent=None, repeat_num=50, T0=0.55, batch_size=256, ratio=0.6):
        self.score_agent, self.energy_agent = score_agent, energy_agent
        self.repeat_num, self.T0, self.batch_size, self.ratio = repeat_num, T0, batch_size, ratio

    def infer_tensors(self, clouds):
        """clouds: device tensor [n,1024,3] -> dict of device tensors with leading dim n (usable under ShardedInference)."""
        n = clouds.shape[0]
        out = {"pred_pose": [], "multi_hypothesis_pred_RTs": [], "energy": [], "sorted_RTs": [], "average_sRT": []}
        for s in range(0, n, self.batch_size):  # evaluation_single.py:380-382 batch slicing
            sample = make_batch_sample(clouds[s:s + self.batch_size])
            pred = self.score_agent.pred_func(data=sample, repeat_num=self.repeat_num, save_path=None, T0=self.T0)
            out["pred_pose"].append(pred)
            out["multi_hypothesis_pred_RTs"].append(rotation.pose9_to_RT(pred))
            if self.energy_agent is not None:
                energy = self.energy_agent.get_energy(data=sample, pose_samples=pred, T=1e-5)  # evaluation_single.py:339-343
                r = reward.rank_aggregate(pred, energy, ratio=self.ratio)
                out["energy"].append(energy)
                out["sorted_RTs"].append(rotation.pose9_to_RT(r["sorted_poses"]))
                out["average_sRT"].append(rotation.quat_trans_to_RT(r["avg_pose"].double()))
        return {k: torch.cat(v, dim=0) for k, v in out.items() if v}

    def infer(self, clouds, device="cuda"):
        """clouds: array-like [n,1024,3].  Returns dict of numpy arrays (+ 'pred_pose' [n,K,9])."""
        clouds = torch.as_tensor(np.asarray(clouds), dtype=torch.float32).to(device)
        return {k: v.cpu().numpy() for k, v in self.infer_tensors(clouds).items()}


    def evaluate(self, detect_result, out_dir=None, degree_thresholds=tuple(range(0, 46)), shift_thresholds=tuple(i / 2 for i in range(21)),
                 iou_thresholds=tuple(i / 100 for i in range(101)), pooling_mode="average", ranker="energy_ranker"):
        """inference_pose -> inference_energy -> evaluate (evaluation_single.py:356-544) on the reference's `detect_result` dict
        (img_path -> {'result', 'valid_pts', 'cat_id', 'valid_inst'}, what detect_mrcnn_genpose pickles): fills every valid
        instance's K hypotheses (ranked by energy, as pred_energy_batch stores them) and energies in place, then computes mAP
        with the reference's threshold grids.  Returns (iou_aps, pose_aps, iou_acc, pose_acc, store)."""
        from . import evaluation
        if self.energy_agent is None:
            raise ValueError("evaluation needs the energy agent (hypotheses are ranked by energy)")
        store = evaluation.DetectionResults(detect_result, self.repeat_num)
        for cat in store.by_category:
            for sl, pts in store.batches(cat, self.batch_size):
                res = self.infer(pts)
                sorted_energy = -np.sort(-res["energy"], axis=1)  # sort_poses_by_energy's second output (reward.py:146-147)
                store.write(cat, sl, res["sorted_RTs"], sorted_energy)
        maps = evaluation.compute_mAP(store.results(), out_dir, list(degree_thresholds), list(shift_thresholds), list(iou_thresholds),
                                      iou_pose_thres=0.1, use_matches_for_pose=True, repeat_num=self.repeat_num, pooling_mode=pooling_mode,
                                      ratio=self.ratio, ranker=ranker)
        return maps + (store,)


# ------------------------------------------------------------------ tracking
def _unit(q):
    return q / q.norm(dim=-1, keepdim=True)


def add_noise_to_RT(RT, r=5.0, t=0.03, draws=None):
    """Initial-pose jitter of the tracking runner (utils/tracking_utils.py:37-101, 'normal' mode): rotate by
    |N(0,1)|*r degrees about a random axis orthogonal (in quaternion space) to the current rotation and shift by
    N(0,1)*t metres along a random direction.  `draws` (tests): the four standard-normal tensors in call order
    (theta [B], quaternion [B,4], shift norm [B], direction [B,3])."""
    B = RT.shape[0]
    if draws is None:
        draws = [torch.randn(B), torch.randn(B, 4), torch.randn(B), torch.randn(B, 3)]
    d_theta, d_q, d_norm, d_dir = [d.to(RT) for d in draws]
    Rm = RT[:, :3, :3]
    theta = (d_theta.abs() * (r / 180 * math.pi)).unsqueeze(-1)
    tr = torch.clamp(1 + Rm[:, 0, 0] + Rm[:, 1, 1] + Rm[:, 2, 2], min=0.0)
    rr = torch.sqrt(tr)
    s = 1.0 / (2 * rr + 1e-7)
    q = _unit(torch.stack((0.5 * rr, (Rm[:, 2, 1] - Rm[:, 1, 2]) * s, (Rm[:, 0, 2] - Rm[:, 2, 0]) * s, (Rm[:, 1, 0] - Rm[:, 0, 1]) * s), dim=-1))
    nq = _unit(d_q)
    q_orth = _unit(nq - q * torch.sum(q * nq, dim=-1, keepdim=True))
    jq = q * torch.cos(theta / 2) + q_orth * torch.sin(theta / 2)
    w, x, y, z = torch.unbind(jq, dim=-1)
    new_R = torch.stack((1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
                         2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
                         2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y), dim=-1).reshape(B, 3, 3)
    out = RT.clone()
    out[:, :3, :3] = new_R
    direction = d_dir / torch.clamp(d_dir.norm(dim=-1, keepdim=True), min=1e-9)
    out[:, :3, 3] = RT[:, :3, 3] + direction * (d_norm * t).unsqueeze(-1)
    return out


def _unpin_frame_entry(key, ent):
    for ws in ent[4]:
        ws["_pins"] -= 1


class _FrameGraphs:
    """hipGraphs of the launch sequences of a tracking frame that need no host decision, keyed by the clouds' shape:
      A   clouds [n,1024,3] -> centres, the SCORE model's per-cloud embedding: grouping (furthest point sampling with its deeper levels on a
          side branch under level 0, ball queries) once, the score model's encoder, its cloud embedding;
      A'  the ENERGY model's encoder on the same centres and neighbourhoods + its cloud embedding.  Nothing before the ranking needs it,
          and the adaptive solve of a frame occupies 10-19 of the 256 CUs: A' is replayed on a side stream and runs UNDERNEATH the solve;
      B   candidates [n,K,9] -> energies -> ranking -> top-`sel` aggregation -> 4x4 poses (energy evaluation, gp_rank_aggregate and
          the small conversions); waits for A'.
    Between A and B sits the adaptive ODE solve, whose attempts are graph replays of their own.  A 5-object frame is then A + the
    solve's replay(s) + B instead of ~70 individually launched kernels.  Inputs are copied into static buffers, outputs are static
    tensors that the NEXT frame of the same shape overwrites (callers that keep them clone)."""

    def __init__(self, snet, enet, K, sel, T_energy=1e-5):
        from .sde import SIGMA_MAX, SIGMA_MIN
        self.snet, self.enet, self.K, self.sel = snet, enet, K, sel
        dev = snet.device
        t = torch.full((1,), float(T_energy), device=dev)
        self.tvec_e = enet.pose_score_net.time_embed(t)[0].contiguous()
        self.sigma_e = (SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** t).contiguous()
        self.share = snet.pts_encoder.grouping_key() == enet.pts_encoder.grouping_key()
        self.side = torch.cuda.Stream(dev)
        self.ev_a, self.ev_e = torch.cuda.Event(), torch.cuda.Event()
        # per object count of a frame: bounded (lru.py); a dropped entry gives its encoder workspaces back to their eviction order
        self._a = ShapeCache(self.MAX_SHAPES, on_evict=_unpin_frame_entry)
        self._b = ShapeCache(self.MAX_SHAPES)

    MAX_SHAPES = 8

    SLOT = "frame-graphs"  # the replays write into encoder workspaces of their own: no ticket of the agent path ever points at them

    def _score_body(self, pts):
        enc_s = self.snet.pts_encoder
        # (deferred join: the deeper sampling levels run on a side branch of the graph underneath the level-0 set abstraction)
        grouping = enc_s.prepare_grouping(pts, slot=self.SLOT, defer_join=True) if self.share else None
        cvec_s = self.snet.pose_score_net.cloud_embed(enc_s.forward(pts, slot=self.SLOT, grouping=grouping))
        return grouping, pts.mean(dim=1), cvec_s

    def _energy_body(self, pts, grouping):
        return self.enet.pose_score_net.cloud_embed(self.enet.pts_encoder.forward(pts, slot=self.SLOT, grouping=grouping))

    def embed(self, pts):
        """-> (centre [n,3], cvec of the score model [n,768], cvec of the energy model [n,768]); the energy model's embedding is still
        being computed on the side stream when this returns - rank() waits for it."""
        key = (tuple(pts.shape), pts.dtype)
        ent = self._a.get(key)
        cur = torch.cuda.current_stream(pts.device)
        if ent is None:
            buf = pts.clone()
            grouping, _, _ = self._score_body(buf)  # warm-up outside capture: workspaces, kernel attributes
            self._energy_body(buf, grouping)
            torch.cuda.synchronize()
            # the replays write into encoder workspaces that were allocated outside the captures (by the warm-up above): pinned - BEFORE the
            # captures, so that none of them can be re-allocated inside one - for as long as these graphs live
            n, N = int(buf.shape[0]), int(buf.shape[1])
            pinned = [self.snet.pts_encoder.pin_workspaces(n, N, self.SLOT), self.enet.pts_encoder.pin_workspaces(n, N, self.SLOT)]
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                grouping, centre, cvec_s = self._score_body(buf)
            ge = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ge, stream=self.side):
                cvec_e = self._energy_body(buf, grouping)
            ent = self._a[key] = (ga, ge, buf, (centre, cvec_s, cvec_e), pinned)
        ga, ge, buf, outs, _ = ent
        cur.wait_event(self.ev_e)  # the previous frame's A' (side stream) has finished reading `buf`, which is about to change
        buf.copy_(pts)
        ga.replay()
        self.ev_a.record(cur)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_a)
            ge.replay()
            self.ev_e.record(self.side)
        return outs

    def _rank_body(self, pred, centre, cvec_e):
        n, K = pred.shape[0], self.K
        pose = pred.to(torch.float32, copy=True)
        pose[:, :, 6:] -= centre.unsqueeze(1)  # posenet_agent.py:516: translations relative to the cloud centre
        energy = self.enet.pose_score_net.evaluate(cvec_e, K, pose.reshape(n * K, 9), self.tvec_e, self.sigma_e, "energy").reshape(n, K, 2)
        r = reward.rank_aggregate(pred, energy, selected_num=self.sel, with_rt=True)  # ranking, aggregation and both 4x4 forms: one launch
        return energy, r["sorted_RTs"], r["avg_RT"]

    def rank(self, pred, centre, cvec_e):
        """pred [n,K,9] f64 -> (energy [n,K,2], sorted_RTs [n,K,4,4], average_sRT [n,4,4])"""
        key = (tuple(pred.shape), pred.dtype)
        ent = self._b.get(key)
        torch.cuda.current_stream(pred.device).wait_event(self.ev_e)  # the energy model's embedding (side stream, under the solve)
        if ent is None:
            bufs = (pred.clone(), centre.clone(), cvec_e.clone())
            self._rank_body(*bufs)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = self._rank_body(*bufs)
            ent = self._b[key] = (g, bufs, outs)
        g, bufs, outs = ent
        for b, v in zip(bufs, (pred, centre, cvec_e)):
            b.copy_(v)
        g.replay()
        return outs


class TrackingRunner:
    """Frame-by-frame tracking with warm-started candidates (main_tracking, evaluation_tracking.py:262-337):
    every instance of a frame forms one batch; the initial pose of an object is the previous frame's aggregated
    sRT for the same `model_name`, else a jittered ground-truth pose; the ODE sampler starts at T0 = 0.15 from
    init_x + prior(T0) (samplers.py:180).

    use_graphs (default): the launch sequences around the adaptive solve are replayed as two hipGraphs per object count
    (_FrameGraphs) - same kernels, same results as the agents' pred_func -> get_energy -> rank_aggregate called one after the other
    (use_graphs=False), a third fewer microseconds per frame at tracking sizes, where launch overhead dominates."""

    def __init__(self, score_agent, energy_agent, repeat_num=50, T0=0.15, ratio=0.6, use_graphs=True):
        self.score_agent, self.energy_agent = score_agent, energy_agent
        self.repeat_num, self.T0, self.ratio = repeat_num, T0, ratio
        self.buffer = {"model_name": [], "pred_sRT": None}
        self.use_graphs = use_graphs
        self._graphs = None

    def reset(self):
        self.buffer = {"model_name": [], "pred_sRT": None}

    def step(self, pts, model_names, gt_RT, noise_draws=None):
        """pts [n,1024,3] device tensor; gt_RT [n,4,4] (used only for objects not seen in the previous frame)."""
        pts = pts.float()
        dev = pts.device
        K = self.repeat_num
        sel = max(1, int(self.ratio * K))
        # (a coupled agent - its batch sharded over a process group - keeps the agent path: the frame graphs build an uncoupled solver)
        graphs = self.use_graphs and self.score_agent.cfg.sampler_mode[0] == "ode" and getattr(self.score_agent.net, "coupling_group", None) is None
        if graphs:
            net = self.score_agent.net
            net._need_weights()
            self.energy_agent.net._need_weights()
            if self._graphs is None:
                self._graphs = _FrameGraphs(net, self.energy_agent.net, K, sel)
            # graph A first: the clouds' centres come out of it (the mean the sample dict would hold, evaluation_tracking.py:306-311)
            centre, cvec_s, cvec_e = self._graphs.embed(pts)
            sample = None
        else:
            sample = make_batch_sample(pts)
            centre = sample["pts_center"]
        with _one_cpu_thread():
            noised = add_noise_to_RT(gt_RT.float().cpu(), draws=noise_draws)  # drawn every frame (:302), whether or not it is used
        if (self.buffer["pred_sRT"] is not None and list(model_names) == self.buffer["model_name"]
                and len(set(model_names)) == len(model_names)):
            # every object continues from the previous frame: nothing to upload, one tensor.  (Unique names only: the reference looks a
            # name up with list.index (evaluation_tracking.py:303-307) - two objects with the SAME name both get the first one's pose,
            # which the general branch below reproduces.)
            init_sRT = self.buffer["pred_sRT"].float()
        else:
            init_sRT = noised.to(dev)
            for i, name in enumerate(model_names):
                if name in self.buffer["model_name"]:
                    init_sRT[i] = self.buffer["pred_sRT"][self.buffer["model_name"].index(name)]
        init_x = torch.cat([init_sRT[:, :3, 0], init_sRT[:, :3, 1], init_sRT[:, :3, 3] - centre], dim=1)  # [R[:,0], R[:,1], t - centre]
        if graphs:
            from .samplers import ODESampler
            n = pts.shape[0]
            prior = net._prior_to_device((n * K, 9), T=self.T0)
            x0 = (prior.view(n, K, 9) + init_x.float().unsqueeze(1)).view(n * K, 9)  # samplers.py:180: init_x repeated K times + prior
            key = ("ode", n, K, None)
            smp = net._samplers.get(key)
            if smp is None:
                smp = net._samplers[key] = ODESampler(net.pose_score_net, n, K, net.device)
            net.last_sampler = smp
            _, x = smp.run(cvec_s, centre, x0, self.T0, num_steps=net.cfg.sampling_steps, eps=net.sampling_eps)
            pred = x.reshape(n, K, 9)
            energy, sorted_RTs, average_sRT = self._graphs.rank(pred, centre, cvec_e)
            energy, sorted_RTs, average_sRT = energy.clone(), sorted_RTs.clone(), average_sRT.clone()
        else:
            pred = self.score_agent.pred_func(data=sample, repeat_num=K, save_path=None, init_x=init_x.float(), T0=self.T0)
            energy = self.energy_agent.get_energy(data=sample, pose_samples=pred, T=1e-5)
            r = reward.rank_aggregate(pred, energy, selected_num=sel, with_rt=True)
            average_sRT, sorted_RTs = r["avg_RT"], r["sorted_RTs"]
        self.buffer = {"model_name": list(model_names), "pred_sRT": average_sRT}
        # (the caller gets its own copy of the aggregated poses: the buffer's tensor is the next frame's warm start)
        return {"init_x": init_x, "pred_pose": pred, "energy": energy, "sorted_RTs": sorted_RTs, "average_sRT": average_sRT.clone()}


class MultiSequenceTracker:
    """Tracking (main_tracking, evaluation_tracking.py:262-337) for SEVERAL sequences at once - BASELINE configs[4]: a frame of one
    sequence is a tiny batch (4-6 objects x 50 candidates), so the frames that different sequences are at share every launch:
    one encoder pass over all their clouds, one device-resident RK45 solve in which every sequence's frame is a group with
    its own step control (gp_rk45_phase_ragged - exactly what a per-sequence cond_ode_sampler call does), one energy pass, one
    ranking launch.  Per sequence the semantics are TrackingRunner's: warm start from the previous frame's aggregated pose
    of the same `model_name`, else the jittered ground truth; T0 = 0.15."""

    def __init__(self, score_agent, energy_agent, n_sequences, repeat_num=50, T0=0.15, ratio=0.6, max_objects_per_frame=8):
        self.score_agent, self.energy_agent = score_agent, energy_agent
        self.repeat_num, self.T0, self.ratio = repeat_num, T0, ratio
        self.buffers = [{"model_name": [], "pred_sRT": None} for _ in range(n_sequences)]
        # one solver sized for the capacity; the group tables are re-filled every step, so frames whose object counts change
        # keep replaying the same captured graphs
        self.cap_groups, self.cap_clouds = n_sequences, n_sequences * max_objects_per_frame
        self._sampler = None
        self._prev = None  # (live sequences, their aggregated poses [objects,4,4] in step order) of the previous step
        self.one_tensor_warm_starts = 0  # steps whose initial poses were the previous step's aggregated poses as they stood

    def reset(self, seq=None):
        for i in (range(len(self.buffers)) if seq is None else [seq]):
            self.buffers[i] = {"model_name": [], "pred_sRT": None}
        self._prev = None

    def step(self, frames, noise_draws=None, prior=None):
        """frames: one (pts [n_i,1024,3] device, model_names [n_i], gt_RT [n_i,4,4]) per sequence (None = the sequence has no
        frame this step).  Returns one TrackingRunner-style dict per sequence (None where there was no frame).
        prior (tests): per sequence, the prior draw [n_i*K,9] exactly as `prior_fn((n_i*K, 9), T=T0)` returns it (already scaled by
        sigma(T0)) - it stands in for prior_fn, unlike the `prior_noise` of the pipeline predictors (standard-normal draws).
        The returned tensors are slices of the step's own result tensors; `average_sRT` is a copy (the tracker keeps the original as the
        next step's warm start, so an in-place edit by the caller - a unit conversion, a scale - cannot reach it)."""
        from .samplers import ODESampler
        net = self.score_agent.net
        net._need_weights()
        K = self.repeat_num
        live = [i for i, f in enumerate(frames) if f is not None and f[0].shape[0] > 0]
        out = [None] * len(frames)
        if not live:
            return out
        dev = frames[live[0]][0].device
        counts = [int(frames[i][0].shape[0]) for i in live]
        pts = torch.cat([frames[i][0].float() for i in live], dim=0)
        centre = pts.mean(dim=1)
        # initial poses: jittered ground truth for every object (drawn every frame, evaluation_tracking.py:302) in ONE host call and
        # one upload, then the warm starts gathered from the previous frames' aggregated poses in one indexed copy
        gt_all = torch.cat([frames[i][2].float().cpu() for i in live], dim=0)
        draws_all = None if noise_draws is None else [torch.cat([noise_draws[i][d] for i in live], dim=0) for d in range(4)]
        with _one_cpu_thread():
            noised = add_noise_to_RT(gt_all, draws=draws_all)  # drawn every frame (evaluation_tracking.py:302), whether or not it is used
        if (self._prev is not None and self._prev[0] == live
                and all(list(frames[i][1]) == self.buffers[i]["model_name"] and len(set(frames[i][1])) == len(frames[i][1]) for i in live)):
            # (unique names per frame only: duplicates go through list.index below, first match, as evaluation_tracking.py:303-307)
            # every object of every sequence continues from the previous step, in the same order: the previous aggregated poses ARE the
            # initial poses - nothing to upload, no index tensors (their pageable host-to-device copies wait for the stream)
            init_sRT = self._prev[1]
            self.one_tensor_warm_starts += 1
        else:
            init_sRT = noised.to(dev)
            prev, src, dst, off, row = [], [], [], 0, 0
            for q, i in enumerate(live):
                buf = self.buffers[i]
                if buf["pred_sRT"] is not None:
                    for j, name in enumerate(frames[i][1]):
                        if name in buf["model_name"]:
                            src.append(off + buf["model_name"].index(name))
                            dst.append(row + j)
                    prev.append(buf["pred_sRT"])
                    off += buf["pred_sRT"].shape[0]
                row += counts[q]
            if src:
                init_sRT[torch.as_tensor(dst, device=dev)] = torch.cat(prev, dim=0)[torch.as_tensor(src, device=dev)].to(init_sRT.dtype)
        init_x = torch.cat([init_sRT[:, :3, 0], init_sRT[:, :3, 1], init_sRT[:, :3, 3] - centre], dim=1)  # [R[:,0], R[:,1], t - centre]
        # ---- score model: encoder -> warm-started ODE, one group per sequence
        shared = {"pts": pts, "pts_center": centre}
        # (launch by launch: the cloud count of a multi-sequence step changes from frame to frame and the pass is not launch-bound here)
        feat = net.extract_pts_feature(shared, use_graph=False)  # leaves the grouping ticket for the energy agent
        cvec = net.pose_score_net.cloud_embed(feat)
        B = pts.shape[0]
        if prior is None:
            pr = net._prior_to_device((B * K, 9), T=self.T0)
        else:
            pr = torch.cat([prior[i].reshape(-1, 9) for i in live], dim=0).to(dev).float()
        x0 = (pr.view(B, K, 9) + init_x.float().unsqueeze(1)).view(B * K, 9)  # samplers.py:180: init_x repeated K times + prior
        if B > self.cap_clouds:
            raise ValueError(f"{B} objects in this step exceed the capacity {self.cap_clouds} (n_sequences x max_objects_per_frame)")
        smp = self._sampler
        if smp is None:
            per = self.cap_clouds // self.cap_groups
            smp = self._sampler = ODESampler(net.pose_score_net, self.cap_clouds, K, dev, group_clouds=[per] * self.cap_groups)
        smp.set_groups(counts)
        _, x = smp.run(cvec, centre, x0, self.T0, num_steps=net.cfg.sampling_steps, eps=net.sampling_eps)
        pred = x.reshape(B, K, 9)
        # ---- energy model + ranking + aggregation for all clouds at once (row / cloud local)
        energy = self.energy_agent.get_energy(data=shared, pose_samples=pred, T=1e-5)
        sel = max(1, int(self.ratio * K))
        r = reward.rank_aggregate(pred, energy, selected_num=sel, with_rt=True)
        average_sRT, sorted_RTs = r["avg_RT"], r["sorted_RTs"]
        lo = 0
        for_caller = average_sRT.clone()  # one small device-side copy, no synchronisation
        for q, i in enumerate(live):
            sl = slice(lo, lo + counts[q])
            lo += counts[q]
            self.buffers[i] = {"model_name": list(frames[i][1]), "pred_sRT": average_sRT[sl]}  # (views of this step's own tensor)
            out[i] = {"init_x": init_x[sl], "pred_pose": pred[sl], "energy": energy[sl], "sorted_RTs": sorted_RTs[sl], "average_sRT": for_caller[sl],
                      "nfev": int(smp.group_stats[q]["nfev"])}
        self._prev = (live, average_sRT)
        return out
