"""Samplers on the HIP kernels: cond_pc_sampler / cond_ode_sampler of networks/gf_algorithms/samplers.py:102-227
(pose_mode 'rot_matrix', VE SDE).  Host code only builds schedule tables, owns buffers and replays hipGraphs;
every per-step operation runs in csrc/scorenet.hip / csrc/rk45.hip.
"""
import numpy as np
import torch

from . import _lib
from ._lib import ptr, stream_ptr
from .sde import EPS, SIGMA_MAX, SIGMA_MIN, ve_sde


def pc_schedule(num_steps, eps=EPS):
    """Host schedule table of the PC sampler: [num_steps,4] = sigma(t_i), g(t_i), step_size, sqrt(step_size),
    evaluated with the reference's own f32 tensor expressions (samplers.py:118-119,145; sde.py:15-24)."""
    time_steps = torch.linspace(1.0, eps, num_steps)
    step_size = time_steps[0] - time_steps[1]
    bt = time_steps.reshape(-1, 1)
    sigma = SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** bt
    _, g = ve_sde(bt)
    sched = torch.cat([sigma, g, step_size.expand(num_steps, 1), torch.sqrt(step_size).expand(num_steps, 1)], dim=1)
    return time_steps, sched.float().contiguous()


class PCSampler:
    """Predictor-corrector sampler state for a fixed (B, K, num_steps): buffers + optional hipGraph of the whole loop."""

    def __init__(self, net, B, K, num_steps, device, use_graph=True, record_traj=False, groups=1, coupling_group=None, tile=0, model="score",
                 precision="f32"):
        """B clouds in `groups` independent batches of B/groups clouds laid out back to back: one launch chain serves all of
        them, the batch-mean gradient norm (samplers.py:130-132) stays per batch (gp_pc_step_grouped).

        coupling_group (a torch.distributed process group; "faithful" multi-GPU mode, SURVEY §8e caveat): this rank's B clouds are
        one SHARD of a batch that is spread over the ranks of the group (equal shards).  After every step the per-batch sums of
        |score| are all-reduced (one float per batch) and the next launch takes the mean over ALL rows of the batch
        (gp_pc_step_plan with gn_ext), so every shard steps exactly as the unsharded batch would.  On RCCL (backend 'nccl') the
        per-step reduction - one device-side sum and one all-reduce of `groups` floats - is captured INSIDE the sampler's hipGraph
        with the launches; on gloo (CPU tests, two ranks sharing one device) the loop runs launch by launch."""
        if B % groups:
            raise ValueError(f"{B} clouds do not split into {groups} equal batches")
        if model not in ("score", "energy"):
            raise ValueError(model)
        # precision 'bf16x3' (OPT-IN, exploratory; csrc/trunk_bf16x3.hip): the trunk's dense layers as three-term bf16 split products with
        # fp32 accumulation, 128-row workgroups.  Score model, no cross-rank coupling; never the default.
        if precision not in ("f32", "bf16x3"):
            raise ValueError(f"sampler precision {precision!r}: 'f32' or 'bf16x3'")
        if precision == "bf16x3" and (model != "score" or coupling_group is not None or tile):
            raise NotImplementedError("the split-bf16 PC step serves the score model, uncoupled, on its own launch plan")
        self.precision = precision
        # model 'energy': `net` holds the ENERGY network's weights and the sampler is driven by ITS score - the gradient of the
        # inner-product energy (posenet.py:94-130 on a PoseEnergyNet), evaluated inside the step kernel (forward + vector-Jacobian product)
        self.model = 0 if model == "score" else 1
        self.net, self.B, self.K, self.n, self.groups = net, B, K, num_steps, groups
        self.dev = torch.device(device)
        R = B * K
        self.R = R
        # launch plan (csrc/score_trunk.h: score_plan_rows): 16 / 32 / 64 = tile form, 128 = chain form (register-resident trunk, weights
        # through an LDS ring) for launches of ~32 000 rows and more; `tile` forces one (tests, measurements)
        import ctypes
        t_out, n_out = ctypes.c_int(0), ctypes.c_int(0)
        if precision == "bf16x3":
            if _lib.lib().gp_pc_layout_bf16x3(groups, B // groups, K, ctypes.byref(n_out)) != 0:
                raise ValueError(f"{B // groups} clouds x {K} candidates per batch do not split into 128-row workgroups of at most four clouds")
            t_out.value = 128
            self._bf = net.w.bf16x3_packs()
        elif _lib.lib().gp_pc_layout(self.model, int(tile), groups, B // groups, K, ctypes.byref(t_out), ctypes.byref(n_out)) != 0:
            raise ValueError(f"{B // groups} clouds x {K} candidates per batch do not split into workgroups of plan {tile or 'auto'}; "
                             "run the batches separately")
        if coupling_group is not None and (t_out.value & _lib.PLAN_HEADSPLIT):
            # a sharded batch: the per-step sums below run over whole-tile partials - the plain 16-row tiles, not the head-split plan
            if _lib.lib().gp_pc_layout(self.model, 16, groups, B // groups, K, ctypes.byref(t_out), ctypes.byref(n_out)) != 0:
                raise ValueError("16-row tiles do not fit this batch")
        # self.plan is what the launches are given; self.tile the rows per workgroup of it; self.hsplit = 3 under the head-split plan of
        # the latency regime (three workgroups per 16-row tile, one head of the network each - GP_PLAN_HEADSPLIT)
        self.plan, self.nparts = t_out.value, n_out.value
        self.tile, self.hsplit = self.plan & ~_lib.PLAN_HEADSPLIT, (3 if self.plan & _lib.PLAN_HEADSPLIT else 1)
        if precision == "bf16x3":
            self.kernel_name = "pc_step_bf16x3_kernel"
        elif self.tile in (16, 32, 64):
            self.kernel_name = ("pc_step_kernel<16,0,split>" if self.hsplit == 3 else f"pc_step_kernel<{self.tile}>" if self.model == 0
                                else f"pc_step_kernel<{self.tile},energy>")
        else:
            self.kernel_name = "pc_step_chain_kernel<2>" if self.model == 0 else "pc_step_chain_kernel<2,energy>"
        ts, sched = pc_schedule(num_steps)
        self.sched = sched.to(self.dev)
        self.tvec_all = net.time_embed(ts.to(self.dev))
        f = lambda *s: torch.empty(*s, device=self.dev)
        self.x, self.mean_x, self.score = f(R, 9), f(R, 9), f(R, 9)
        self.partials = torch.zeros(num_steps, self.nparts, device=self.dev)
        self.z1, self.z2 = f(num_steps, R, 9), f(num_steps, R, 9)
        self.cvec, self.centre = f(B, 768), f(B, 3)
        self.traj = f(num_steps, R, 9) if record_traj else None
        self.coupling_group = coupling_group
        self.gn_ext, self.gn_rows = None, 0
        if coupling_group is not None:
            import torch.distributed as dist
            self._dist = dist
            self._world = dist.get_world_size(coupling_group)
            self.gn_ext = torch.zeros(num_steps, groups, device=self.dev)
            self.gn_rows = R // groups * self._world  # rows of a batch over all its (equal) shards
            use_graph = use_graph and dist.get_backend(coupling_group) == "nccl"  # an RCCL all-reduce is graph-capturable, a gloo one is not
        self.use_graph = use_graph
        self.graph = None

    def launch_step(self, i):
        """Launch i of the chain (0 <= i <= n) on the current stream: finishes step i-1 and, for i < n, evaluates the score at t_i."""
        if self.precision == "bf16x3":
            t = self.net.w.tensors
            _lib.call("gp_pc_step_bf16x3", self.groups, self.B // self.groups, self.K, i, self.n, ptr(self.cvec), ptr(self.tvec_all), ptr(self.sched),
                      ptr(self.z1), ptr(self.z2), ptr(self.centre), ptr(self.x), ptr(self.mean_x), ptr(self.score), ptr(self.partials), ptr(self.traj),
                      ptr(self._bf[0]), ptr(self._bf[1]), ptr(self._bf[2]), ptr(t["b_pose0"]), ptr(t["b_pose2"]), ptr(t["w_out"]), ptr(t["b_out"]),
                      stream_ptr())
            return
        _lib.call("gp_pc_step_plan", self.model, self.plan, self.groups, self.B // self.groups, self.K, i, self.n, self.net.w.ref(), ptr(self.cvec), ptr(self.tvec_all),
                  ptr(self.sched), ptr(self.z1), ptr(self.z2), ptr(self.centre), ptr(self.x), ptr(self.mean_x), ptr(self.score), ptr(self.partials),
                  ptr(self.traj), ptr(self.gn_ext), self.gn_rows, stream_ptr())

    def _launch_all(self):
        for i in range(self.n + 1):
            self.launch_step(i)
            if self.coupling_group is not None and i < self.n:
                # sum of |score_i| over this shard's rows, per batch -> over all shards (the kernel divides by the batch's row count)
                torch.sum(self.partials[i].view(self.groups, -1), dim=1, out=self.gn_ext[i])
                self._dist.all_reduce(self.gn_ext[i], op=self._dist.ReduceOp.SUM, group=self.coupling_group)

    def run(self, cvec, centre, init_x, z_langevin=None, z_predictor=None, slot_free_event=None, graph_events=None):
        """cvec [B,768], centre [B,3], init_x [R,9]; noise [n,R,9] (drawn on the device generator if None).
        Returns (xs [R,n,9] or None, mean_x [R,9]) float32, like cond_pc_sampler.
        slot_free_event: recorded once the inputs have been copied into the sampler's own buffers (pipelining)."""
        self.cvec.copy_(cvec)
        self.centre.copy_(centre)
        self.x.copy_(init_x)
        if slot_free_event is not None:
            slot_free_event.record(torch.cuda.current_stream())
        if z_langevin is None:
            self.z1.normal_()
            self.z2.normal_()
        else:
            self.z1.copy_(z_langevin)
            self.z2.copy_(z_predictor)
        if not self.use_graph:
            self._launch_all()
        else:
            if self.graph is None:
                # warm-up launch outside capture (sets kernel attributes), then capture the whole T-step loop once
                x0 = self.x.clone()
                self._launch_all()
                torch.cuda.synchronize()
                self.x.copy_(x0)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self._launch_all()
            if graph_events is not None:
                graph_events[0].record(torch.cuda.current_stream())
            self.graph.replay()
            if graph_events is not None:
                graph_events[1].record(torch.cuda.current_stream())
        xs = self.traj.permute(1, 0, 2) if self.traj is not None else None
        return xs, self.mean_x


# ---------------------------------------------------------------------------------------------- PF-ODE (RK45)
_STATE_FIELDS = ("t", "h_abs", "status", "n_attempts", "n_accepted", "nfev", "err_norm", "log_t", "log_h", "log_err", "log_acc",
                 "stage_t", "last_accepted")


def _state_layout():
    import ctypes
    arr = (ctypes.c_int64 * 16)()
    _lib.call("gp_rk45_state_layout", arr, 16)
    return {n: int(arr[i]) for i, n in enumerate(_STATE_FIELDS)}, int(_lib.lib().gp_rk45_state_bytes())


class ODESampler:
    """cond_ode_sampler (samplers.py:163-227) with the whole Dormand-Prince loop resident on the GPU.

    One *attempt* (stage-time embedding + 6 fused stage kernels + controller) is captured as a hipGraph and replayed;
    the host only polls the device-side `status` word every `poll` attempts (the number of attempts is data dependent:
    scipy's adaptive controller, rtol = atol = 1e-5, batch-global RMS error norm)."""

    TRAJ_CAP = 192
    CHUNKS = (8, 12, 16, 24, 32, 40, 48, 64, 80, 96, 128)  # attempts per first replay

    MAX_GRAPHS = 12  # captured attempt graphs kept per kind (each holds chunk x 8 kernel nodes)
    MODELS = {"score": 0, "energy": 1, "likelihood": 2}

    def __init__(self, net, B, K, device, use_graph=True, poll=8, groups=1, group_clouds=None, model="score", coupling_group=None, tile=0):
        """B clouds in `groups` independent batches of B/groups clouds laid out back to back: every batch keeps its own adaptive
        step control (error norm over ITS rows, accept / reject, step size - what separate cond_ode_sampler calls would do) while
        all of them share each launch (gp_rk45_phase_grouped).

        model: what the driver integrates (gp_rk45_phase_model) - 'score' the probability-flow ODE of the score network; 'energy' the
        same ODE with the ENERGY network's score (`net` holds its weights; forward + vector-Jacobian product inside the stage
        kernels); 'likelihood' the [pose, log-density] ODE of cond_ode_likelihood (run_likelihood).

        coupling_group (a torch.distributed process group; "faithful" multi-GPU mode, SURVEY §8e caveat): this rank's B clouds are one
        SHARD of a batch spread over the ranks of the group (equal shards).  scipy's error norm - and the norms of its initial-step
        heuristic - run over the WHOLE batch: after the stage kernels the per-group sums of squares are all-reduced (two doubles per
        group) and the step controller decides on the reduced sums, so every shard takes the accept / reject sequence of the unsharded
        batch.  On RCCL the all-reduce is captured with the attempts; on gloo the attempts run launch by launch.

        tile: launch plan of the stage kernels (0 = pick: 16- / 32-row tiles, or the 128-row chain form of the trunk for score-model
        launches of ~32 000 rows and more, gp_rk45_plan_rows); tests and measurements force one."""
        self.model = self.MODELS[model]
        self.ncomp = 10 if self.model == 2 else 9
        self.ragged = group_clouds is not None
        self.shared = False
        if self.ragged and self.model != 0:
            raise NotImplementedError("ragged groups integrate the score network's ODE only")
        self.dev = torch.device(device)
        if self.ragged:
            # groups of different sizes (tracking: the objects of one frame): consecutive cloud ranges, tiles that end at the group
            # boundary (gp_rk45_phase_ragged).  B and len(group_clouds) are CAPACITIES: set_groups() re-fills the device tables for
            # any grouping that fits, so the captured graphs (fixed grids) serve frames whose object counts change.
            groups = len(group_clouds)
            self.tile = (int(tile) & ~_lib.PLAN_HEADSPLIT) if tile else _lib.lib().gp_score_tile_rows(B * K)  # 16 rows, or 32 once the launch is MFMA-bound
            if self.tile not in (16, 32):
                raise ValueError("ragged groups run on 16- or 32-row tiles")
            self.nblocks = (B * K + self.tile - 1) // self.tile + groups  # every group wastes less than one tile
            # latency regime (one sequence's frames: a few dozen tiles at capacity): three workgroups per tile, one head of the network each
            if tile:
                self.hsplit = 3 if int(tile) & _lib.PLAN_HEADSPLIT else 1
            else:
                self.hsplit = 3 if self.tile == 16 and _lib.lib().gp_plan_headsplit_pays(self.nblocks) else 1
            self.plan = self.tile | (_lib.PLAN_HEADSPLIT if self.hsplit == 3 else 0)
            self.blk_info = torch.zeros(self.nblocks, 3, dtype=torch.int32, device=self.dev)
            self.grp_info = torch.zeros(groups, 4, dtype=torch.int32, device=self.dev)
            self._tables_host = (torch.zeros(self.nblocks, 3, dtype=torch.int32).pin_memory(), torch.zeros(groups, 4, dtype=torch.int32).pin_memory())
        elif B % groups:
            raise ValueError(f"{B} clouds do not split into {groups} equal batches")
        self.net, self.B, self.K, self.groups = net, B, K, groups
        R = self.R = B * K
        if not self.ragged:
            # forward + backward right-hand sides (energy model, likelihood): 16-row tiles or, for large launches, the 128-row chain form
            # (a sharded batch's controller runs on all-reduced per-group sums between the stage launches and the decision: not the shared-chunk plan)
            pick = _lib.lib().gp_rk45_plan_rows if coupling_group is None else _lib.lib().gp_rk45_plan_rows_unshared
            self.plan = int(tile) if tile else pick(self.model, groups, B // groups, K)
            # plan = rows per workgroup | GP_PLAN_HEADSPLIT (three workgroups per 16-row tile, one head each - the latency regime)
            #                           | GP_PLAN_SHARED (one workgroup per CU, the left-over 16-row chunks shared across the stages of an attempt)
            self.tile, self.hsplit = self.plan & ~_lib.PLAN_FLAGS, (3 if self.plan & _lib.PLAN_HEADSPLIT else 1)
            self.shared = bool(self.plan & _lib.PLAN_SHARED)
            if ((self.hsplit == 3 and (self.tile != 16 or self.model != 0)) or self.tile not in (16, 32, 48, 64, 128) or (self.tile == 48 and not self.shared)
                    or (self.model != 0 and self.tile in (32, 48, 64)) or (groups > 1 and (R // groups) % self.tile)):
                raise ValueError(f"{B // groups} clouds x {K} candidates per batch do not split into workgroups of plan {tile or 'auto'}; "
                                 "run the batches separately")
            npart = _lib.lib().gp_rk45_partials_count(self.model, self.plan, groups, B // groups, K)
            if npart <= 0:
                raise ValueError(f"plan {self.plan:#x} does not serve {groups} x {B // groups} clouds x {K} candidates")
            self.nblocks = npart // (3 * self.hsplit)
        if self.ragged:
            self.set_groups(group_clouds)
        self.layout, nbytes = _state_layout()
        self.state_bytes = nbytes
        self.state = torch.zeros(groups * nbytes, dtype=torch.uint8, device=self.dev)
        d = lambda *s: torch.zeros(*s, dtype=torch.float64, device=self.dev)
        nc = self.ncomp
        self.y, self.ynew, self.Kbuf = d(R * nc), d(R * nc), d(7, R * nc)
        self.partials = d(3, self.nblocks * self.hsplit)
        self.x_out = d(R, nc)
        self.probe = torch.zeros(R, 9, device=self.dev) if self.model == 2 else None
        self.coupling_group, self.ext_sums, self.ext_rows = coupling_group, None, 0
        if coupling_group is not None:
            if self.ragged:
                raise NotImplementedError("ragged groups are not sharded")
            import torch.distributed as dist
            self._dist = dist
            self.ext_sums = d(2, groups)
            self.ext_rows = R // groups * dist.get_world_size(coupling_group)
            use_graph = use_graph and dist.get_backend(coupling_group) == "nccl"
        self.tvec = torch.zeros(groups * 8, 768, device=self.dev)
        self.cvec = torch.empty(B, 768, device=self.dev)
        self.centre = torch.empty(B, 3, device=self.dev)
        self.traj = None
        if self.model == 2 and poll == 8:
            poll = 64  # the likelihood ODE runs from eps to 1 at rtol 1e-5: thousands of attempts, fewer status reads
        self.use_graph, self.poll = use_graph, poll
        self._graphs = {}        # kind ('graph' | 'graph_traj' | 'graph_dense') -> {attempts per replay: captured graph}
        self._attempt_hist = {}  # (kind, T0) -> attempts the previous solve took
        self.last_stats = {}

    def set_groups(self, group_clouds):
        """Ragged mode: (re)define the groups - clouds per group, consecutive; fewer groups / clouds than the capacity leave padding
        workgroups and padding groups that exit at once.  Returns the number of clouds in use."""
        group_clouds = [int(c) for c in group_clouds]
        if len(group_clouds) > self.groups or sum(group_clouds) > self.B or (group_clouds and min(group_clouds) <= 0):
            raise ValueError(f"groups {group_clouds} exceed the capacity ({self.groups} groups, {self.B} clouds)")
        blk_h, grp_h = self._tables_host
        # the pinned host tables are rewritten below: only the previous upload out of them has to be complete (the device
        # tables themselves are overwritten in stream order, behind every launch that still reads them)
        if getattr(self, "_tables_ev", None) is not None:
            self._tables_ev.synchronize()
        blk, grp = blk_h.numpy(), grp_h.numpy()
        blk[:] = 0
        grp[:] = 0
        if group_clouds:
            rows = np.asarray(group_clouds, dtype=np.int64) * self.K
            nb = (rows + self.tile - 1) // self.tile
            row0 = np.concatenate([[0], np.cumsum(rows)[:-1]])
            blk0 = np.concatenate([[0], np.cumsum(nb)[:-1]])
            ng = len(group_clouds)
            grp[:ng] = np.stack([blk0, nb, rows, row0], axis=1)
            gid = np.repeat(np.arange(ng), nb)
            local = np.arange(int(nb.sum())) - np.repeat(blk0, nb)
            blk[: len(gid)] = np.stack([gid, row0[gid] + local * self.tile, (row0 + rows)[gid]], axis=1)
        self.blk_info.copy_(blk_h, non_blocking=True)
        self.grp_info.copy_(grp_h, non_blocking=True)
        if getattr(self, "_tables_ev", None) is None:
            self._tables_ev = torch.cuda.Event()
        self._tables_ev.record(torch.cuda.current_stream(self.dev))
        self.group_clouds = group_clouds
        return sum(group_clouds)

    def _phase(self, phase, traj=None, t0=0.0, t_bound=0.0, rtol=1e-5, atol=1e-5, dscale=0.0, do_denoise=1, nstates=0):
        import ctypes
        cd = ctypes.c_double
        tail = (ptr(self.cvec), ptr(self.tvec), ptr(self.centre), ptr(self.state), ptr(self.y), ptr(self.ynew), ptr(self.Kbuf), ptr(self.partials),
                ptr(traj), 0 if traj is None else traj.shape[0], cd(t0), cd(t_bound), cd(rtol), cd(atol), cd(dscale), do_denoise, nstates,
                ptr(self.x_out), stream_ptr())
        if self.ragged:
            _lib.call("gp_rk45_phase_ragged", phase, self.groups, ptr(self.grp_info), self.nblocks, ptr(self.blk_info), self.plan, self.B, self.K,
                      self.net.w.ref(), *tail)
        else:
            _lib.call("gp_rk45_phase_model", self.model, self.plan, ptr(self.probe), phase, self.groups, self.B // self.groups, self.K, self.net.w.ref(), *tail[:-1],
                      ptr(self.ext_sums), self.ext_rows, tail[-1])
            if self.ext_sums is not None and phase in (1, 2, 3):
                # sharded batch: the controller decides on the sums of squares over ALL shards
                self._dist.all_reduce(self.ext_sums, op=self._dist.ReduceOp.SUM, group=self.coupling_group)
                _lib.call("gp_rk45_phase_model", self.model, self.plan, ptr(self.probe), phase + 10, self.groups, self.B // self.groups, self.K, self.net.w.ref(),
                          *tail[:-1], ptr(self.ext_sums), self.ext_rows, tail[-1])

    def _attempt(self, traj):
        # the step controller at the end of phase 3 (and of phase 2 before the first attempt) also writes the time embeddings of the
        # next attempt's six stage times: an attempt is six stage launches + one controller launch
        self._phase(3, traj)

    def _read_states(self):
        if getattr(self, "_state_host", None) is None:
            self._state_host = torch.empty(self.state.shape, dtype=torch.uint8).pin_memory()
        self._state_host.copy_(self.state, non_blocking=True)  # one D2H copy for all groups, into pinned memory
        torch.cuda.current_stream(self.dev).synchronize()
        raw_all = self._state_host.numpy()
        return [self._parse_state(raw_all[g * self.state_bytes:(g + 1) * self.state_bytes]) for g in range(self.groups)]

    def _read_state(self, group=0):
        return self._read_states()[group]

    def _parse_state(self, raw):
        L = self.layout
        g = lambda name, dt, n=1: np.frombuffer(raw.tobytes(), dtype=dt, count=n, offset=L[name])
        st = {k: g(k, np.int32)[0] for k in ("status", "n_attempts", "n_accepted", "nfev")}
        st.update({k: g(k, np.float64)[0] for k in ("t", "h_abs", "err_norm")})
        na = min(int(st["n_attempts"]), 512)
        st["log_t"], st["log_h"], st["log_err"] = (g(k, np.float64, 512)[:na].copy() for k in ("log_t", "log_h", "log_err"))
        st["log_acc"] = g("log_acc", np.int32, 512)[:na].copy()
        return st

    def _solve(self, traj, gname, T0, max_attempts=4096):
        """The adaptive loop after phases 0-2: replays captured attempts until every group's device-side status word is set.
        Returns the per-group states."""
        n_done = 0
        # The attempt count is data dependent (scipy's controller), but it barely moves between solves of the same kind (same T0, same
        # kind of clouds), and an attempt launched on a FINISHED solve exits at once.  So the first replay is a graph sized for the
        # previous solve's attempt count (+ margin): in the common case the whole adaptive loop is ONE graph replay and one status
        # read; a solve that needs more continues in chunks of `poll` attempts.
        hist_key = (gname, round(float(T0), 3))
        expect = self._attempt_hist.get(hist_key)
        # (solves of up to 64 attempts - tracking: 6-8, the benched ODE-100: 36 - get a graph of exactly one spare attempt: every attempt
        # launched on a finished solve is eight kernels that exit at once, ~35 us, and a coarse chunk list made 12 attempts out of 7)
        first = self.poll if expect is None else (expect + 1 if expect < 64 else next((c for c in self.CHUNKS if c >= expect + 2), self.CHUNKS[-1]))
        if expect is not None and expect < 64 and self.use_graph:
            # a solve whose attempt count drifts by one or two from frame to frame must not pay a capture (torch.cuda.graph synchronises
            # the device) in the middle of a latency-critical frame: take an already captured graph with up to three more spare attempts
            # (each is eight kernels that exit at once) before capturing a new size
            have = [c for c in self._graphs.get(gname, {}) if first <= c <= first + 3]
            if have:
                first = min(have)
        while True:
            chunk = first if n_done == 0 else self.poll
            if self.use_graph:
                graphs = self._graphs.setdefault(gname, {})
                if chunk not in graphs:
                    if not graphs:
                        self._attempt(traj)  # warm-up outside capture
                        n_done += 1
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for _ in range(chunk):
                            self._attempt(traj)
                    graphs[chunk] = g
                    if len(graphs) > self.MAX_GRAPHS:  # bounded cache: drop the least recently captured size that is not in use now
                        for c in list(graphs):
                            if c not in (chunk, self.poll):
                                del graphs[c]
                                break
                graphs[chunk].replay()
            else:
                for _ in range(chunk):
                    self._attempt(traj)
            n_done += chunk
            sts = self._read_states()
            if self.ragged:
                sts = sts[: len(self.group_clouds)]
            if all(s_["status"] != 0 for s_ in sts):
                break
            if n_done >= max_attempts:
                raise RuntimeError("ODE sampler: attempt budget exhausted")
        if any(s_["status"] == -2 for s_ in sts):
            raise RuntimeError("ODE sampler: a stage of the shared-chunk plan never saw its predecessor's state (bounded wait ran out); "
                               "force a whole-tile plan with ODESampler(..., tile=64)")
        if any(s_["status"] < 0 for s_ in sts):
            raise RuntimeError("ODE sampler: required step size is less than spacing between numbers (scipy TOO_SMALL_STEP)")
        self.group_stats = sts
        self._attempt_hist[hist_key] = max(int(s_["n_attempts"]) for s_ in sts)
        self.last_replays = {"first_chunk": first, "attempts_launched": n_done}
        return sts

    def run_likelihood(self, cvec, x, probe, eps=EPS, rtol=1e-5, atol=1e-5, max_attempts=16384):
        """cond_ode_likelihood's integration (samplers.py:73-93) on the device: state [x, logp] from t = eps to t = 1 with the fixed
        Hutchinson probe.  cvec [B,768]; x, probe [B*K,9].  Returns (z [R,9] f64, delta_logp [R] f64); evaluation count in
        last_stats['nfev'] (2 for the initial step + 6 per attempt, like solve_ivp)."""
        if self.model != 2:
            raise RuntimeError("ODESampler(model='likelihood') required")
        R = self.R
        if cvec.shape[0] != self.B or x.shape[0] != R or probe.shape[0] != R:
            raise ValueError(f"likelihood solver set up for {self.B} clouds x {self.K} rows got {cvec.shape[0]} clouds / {x.shape[0]} rows")
        self.cvec.copy_(cvec)
        self.centre.zero_()
        self.probe.copy_(probe.float())
        y0 = self.y.view(R, 10)
        y0[:, :9].copy_(x.double())   # solve_ivp casts the initial state to float64
        y0[:, 9].zero_()
        self._phase(0, None, t0=eps, t_bound=1.0, rtol=rtol, atol=atol)
        self._phase(1, None)
        self._phase(2, None)
        sts = self._solve(None, "graph", eps, max_attempts)
        self._phase(5, None)
        self.last_stats = sts[0]
        out = self.x_out.clone()
        return out[:, :9], out[:, 9]

    def run(self, cvec, centre, init_x, T0, num_steps=None, eps=EPS, rtol=1e-5, atol=1e-5, denoise=True, return_process=False,
            max_attempts=4096):
        """Returns (xs [R,S,9] f64 or None, x [R,9] f64).  With num_steps=None the in-process samples are the accepted
        states (like solve_ivp without t_eval)."""
        if self.model == 2:
            raise RuntimeError("ODESampler(model='likelihood') integrates the likelihood ODE: call run_likelihood()")
        dense = return_process and num_steps is not None
        if self.groups > 1 and return_process and not dense:
            raise NotImplementedError("accepted-state trajectories have a different length per batch: ask for them one batch at a time")
        nb_in = cvec.shape[0]  # ragged mode may use fewer clouds than the capacity
        expect = sum(self.group_clouds) if self.ragged else self.B
        if nb_in != expect or init_x.shape[0] != nb_in * self.K:
            raise ValueError(f"ODE sampler set up for {expect} clouds x {self.K} candidates got {nb_in} clouds / {init_x.shape[0]} rows"
                             + (" (call set_groups() with this step's grouping first)" if self.ragged else ""))
        self.cvec[:nb_in].copy_(cvec)
        self.centre[:nb_in].copy_(centre)
        self.y[: nb_in * self.K * 9].copy_(init_x.reshape(-1))  # init_x f32 -> f64 state in the copy (solve_ivp casts y0 to float64)
        traj = None
        if dense:
            # solve_ivp(t_eval=np.linspace(T0, eps, num_steps)): 4th-order dense output at every t_eval point
            import ctypes
            from scipy.integrate._ivp.rk import RK45  # published dense-output constants (7x4 matrix P)
            key = ("dense", num_steps)
            if getattr(self, "_dense_key", None) != key:
                self._dense_traj = torch.zeros(num_steps, self.R * 9, dtype=torch.float64, device=self.dev)
                self._dense_key = key
                self._graphs.pop("graph_dense", None)  # the captured attempts hold the trajectory pointer
            self._t_eval = torch.from_numpy(np.linspace(T0, eps, num_steps)).to(self.dev)
            traj = self._dense_traj
            self._phase(0, None, t0=T0, t_bound=eps, rtol=rtol, atol=atol)
            Pm = np.ascontiguousarray(RK45.P, dtype=np.float64)
            _lib.call("gp_rk45_set_dense_grouped", self.groups, ptr(self.state), ptr(self._t_eval), num_steps, Pm.ctypes.data_as(ctypes.c_void_p),
                      stream_ptr())
        else:
            if return_process:
                if self.traj is None:
                    self.traj = torch.zeros(self.TRAJ_CAP, self.R * 9, dtype=torch.float64, device=self.dev)
                traj = self.traj
            self._phase(0, traj, t0=T0, t_bound=eps, rtol=rtol, atol=atol)
        self._phase(1, traj)
        self._phase(2, traj)
        gname = "graph_dense" if dense else ("graph_traj" if traj is not None else "graph")
        sts = self._solve(traj, gname, T0, max_attempts)
        st = sts[0]
        self._phase(4, traj, t0=eps)
        nstates = (num_steps if dense else int(st["n_accepted"]) + 1) if traj is not None else 0
        if traj is not None and not dense and nstates > self.TRAJ_CAP:
            raise RuntimeError(f"ODE sampler: {nstates} accepted states exceed the trajectory capacity {self.TRAJ_CAP}")
        dscale = (1 - eps) / (1000 if num_steps is None else num_steps)
        self._phase(5, traj, dscale=dscale, do_denoise=1 if denoise else 0, nstates=nstates)
        for s_ in sts:
            s_["nfev"] = int(s_["nfev"]) + (1 if denoise else 0)
        self.last_stats = st
        xs = None
        if traj is not None:
            xs = traj[:nstates].reshape(nstates, self.R, 9).permute(1, 0, 2).clone()
        return xs, self.x_out[: nb_in * self.K].clone()
