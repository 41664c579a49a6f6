"""GFObjectPose on the HIP kernels - same constructor arguments and string-dispatched forward as the reference's
networks/posenet.py:18-179, inference modes only:

    net(data, mode='pts_feature')                      -> [B,1024]          (posenet.py:71-91)
    net(data, mode='score' | 'energy')                 -> [R,9] | [R,2]     (posenet.py:159-164)
    net(data, mode='pc_sample' | 'ode_sample', init_x=, T0=) -> (in_process [R,S,9], res [R,9])  (posenet.py:94-130)

`data` is the reference's dict of device tensors ('pts', 'pts_feat', 'sampled_pose', 't', 'pts_center').
Extra (not in the reference): data may carry '_repeat' = K, meaning 'pts_feat'/'pts_center' hold one row per CLOUD
and every cloud owns K consecutive pose rows - this is how PoseNet.pred_func avoids the K-fold repeat of the inputs
(posenet_agent.py:426-434 repeats even the raw clouds: 315 MB at B=256).
"""
import torch

from .encoder import Pointnet2EncoderHIP
from .lru import ShapeCache
from .samplers import ODESampler, PCSampler
from .scorenet import ScoreNetHIP
from .sde import SIGMA_MAX, SIGMA_MIN


class GFObjectPose:
    MAX_SAMPLERS = 8  # distinct (sampler, B, K, steps ...) geometries kept alive
    BF16X3_MIN_ROWS = 9600  # opt-in split-bf16 PC step: launches from here on (profiles/r5_pc_bf16x3_plans.txt: 6 400 rows 47 vs 39 us, 16 000 rows 48 vs 71)

    def __init__(self, cfg, prior_fn, marginal_prob_fn, sde_fn, sampling_eps, T):
        self.cfg = cfg
        self.device = torch.device(cfg.device)
        self.prior_fn, self.marginal_prob_fn, self.sde_fn = prior_fn, marginal_prob_fn, sde_fn
        self.sampling_eps, self.T = sampling_eps, T
        if cfg.pts_encoder != "pointnet2":
            raise NotImplementedError(f"pts_encoder '{cfg.pts_encoder}': only the default PointNet++ encoder is on the MI355X hot path")
        if getattr(cfg, "regression_head", "Rx_Ry_and_T") != "Rx_Ry_and_T" or getattr(cfg, "pose_mode", "rot_matrix") != "rot_matrix":
            raise NotImplementedError("only regression_head='Rx_Ry_and_T' with pose_mode='rot_matrix' is implemented")
        if cfg.posenet_mode not in ("score", "energy"):
            raise NotImplementedError(cfg.posenet_mode)
        if cfg.posenet_mode == "energy":
            for k, v in (("energy_mode", "IP"), ("s_theta_mode", "score"), ("norm_energy", "identical")):
                if getattr(cfg, k, v) != v:
                    raise NotImplementedError(f"{k}='{getattr(cfg, k)}' (only the shipped default '{v}')")
        self.pts_encoder = None
        self.pose_score_net = None
        # samplers (captured launch chains + noise / solver-state buffers) and pinned staging buffers per batch geometry: bounded,
        # least recently used first (lru.py) - a ragged tail per category / a new object count per image must not accumulate
        self._samplers = ShapeCache(self.MAX_SAMPLERS)
        self._staging = ShapeCache(self.MAX_SAMPLERS)
        self.training = False

    # ------------------------------------------------------------------ nn.Module-like surface
    def load_state_dict(self, state_dict, strict=True):
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
        params = getattr(self.cfg, "pointnet2_params", "light")
        self.pts_encoder = Pointnet2EncoderHIP(sd, self.device, params, arith=getattr(self.cfg, "dist_arith", None),
                                               precision=getattr(self.cfg, "encoder_precision", "f32"))
        self.pose_score_net = ScoreNetHIP(sd, self.device)
        self._samplers.clear()
        return self

    def eval(self):
        self.training = False
        return self

    def to(self, device):
        return self

    def _need_weights(self):
        if self.pose_score_net is None:
            raise RuntimeError("GFObjectPose has no weights: call load_state_dict()/PoseNet.load_ckpt() first")

    # ------------------------------------------------------------------ pieces
    def extract_pts_feature(self, data, use_graph=True):
        """posenet.py:71-91.  The sampled centres and ball-query neighbourhoods depend on the coordinates only, so the SCORE agent
        leaves a ticket for them in the dict (`_grouping`) and the ENERGY agent, called next with the same dict and the same clouds
        (evaluation_single.py:339-343, evaluation_tracking.py:316-321), takes them over instead of recomputing them."""
        self._need_weights()
        enc, pts = self.pts_encoder, data["pts"]
        if self.cfg.posenet_mode == "energy" and enc.ticket_valid(data.get("_grouping"), pts, enc.grouping_key()):
            return enc.encode(pts, grouping=data["_grouping"]["ws"], use_graph=use_graph and data["_grouping"].get("use_graph", True))
        feat, ws = enc.encode(pts, use_graph=use_graph)  # one hipGraph replay per input shape from the second call on (encoder.py)
        data["_grouping"] = dict(enc.grouping_ticket(pts, ws), use_graph=use_graph)  # the second agent follows the first one's choice
        return feat

    def _rows(self, data):
        """-> (cvec [B,768], K, centre [B,3] or None)"""
        K = int(data.get("_repeat", 1))
        cvec = self.pose_score_net.cloud_embed(data["pts_feat"].float())
        return cvec, K

    def _prior_to_device(self, shape, **kw):
        """prior_fn draws on the CPU generator like the reference (sde.py:28); the result goes through a cached PINNED staging
        buffer - a pageable host-to-device copy takes the runtime's slow path (staging allocation, occasional multi-ms stalls)."""
        # single-threaded on purpose: a multi-threaded CPU op leaves its OpenMP team spinning next to the HIP runtime's
        # progress thread, and every few calls a sampler graph replay stalls for 60-70 ms (measured at 256 clouds:
        # 12.9 ms per pred_func with one thread, 13 / 13 / 73 ms with the default 128)
        nt = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            return self._prior_to_device_1t(shape, **kw)
        finally:
            torch.set_num_threads(nt)

    def _prior_to_device_1t(self, shape, **kw):
        cpu = self.prior_fn(shape, **kw)
        if cpu.is_cuda:
            return cpu.float()
        key = tuple(cpu.shape)
        st = self._staging.get(key)
        if st is None:
            st = self._staging[key] = (torch.empty(key, dtype=torch.float32).pin_memory(), torch.cuda.Event())
            st[1].record()
        pinned, ev = st
        ev.synchronize()  # the previous copy out of the pinned buffer has completed
        pinned.copy_(cpu)
        dev = pinned.to(self.device, non_blocking=True)
        ev.record()
        return dev

    def _pc_precision(self, B, K, coupling=None, groups=1):
        """cfg.sampler_precision = 'bf16x3' (opt-in, exploratory): the PC sampler's score network on split-bf16 products where the 128-row
        launch plan applies (>= 43 candidates per cloud, uncoupled); anything else keeps the fp32 kernels."""
        if getattr(self.cfg, "sampler_precision", "f32") != "bf16x3" or coupling is not None or K < 43:
            return "f32"
        if B * K < self.BF16X3_MIN_ROWS:  # a round of its 128-row workgroups costs ~47 us whatever the fill: below, the fp32 plans are faster
            return "f32"
        return "bf16x3" if groups == 1 or ((B // groups) * K) % 128 == 0 else "f32"

    def sample(self, data, sampler, init_x=None, T0=None, noise=None, return_process=True):
        self._need_weights()
        cvec, K = self._rows(data)
        B = cvec.shape[0]
        R = B * K
        centre = data["pts_center"].float()
        if self.cfg.posenet_mode == "energy":
            return self._sample_energy_model(cvec, K, centre, sampler, init_x, T0, noise, return_process)
        if sampler == "pc":
            n = self.cfg.sampling_steps
            if n is None:
                raise ValueError("the PC sampler needs cfg.sampling_steps")
            x0 = self._prior_to_device((R, 9)) if init_x is None else init_x.float()
            coupling = getattr(self, "coupling_group", None)
            key = ("pc", B, K, n, return_process, id(coupling) if coupling is not None else None)  # a coupled sampler is a different sampler
            smp = self._samplers.get(key)
            if smp is None:
                # self.coupling_group (optional, set by the caller): the batch is sharded over the ranks of that process group and
                # the Langevin step size is taken over ALL of its rows (PCSampler, "faithful" multi-GPU mode)
                smp = self._samplers[key] = PCSampler(self.pose_score_net, B, K, n, self.device, record_traj=return_process,
                                                      coupling_group=coupling, precision=self._pc_precision(B, K, coupling))
            z1, z2 = noise if noise is not None else (None, None)
            self.last_sampler = smp  # statistics / timing of the sampler that served the last call
            xs, res = smp.run(cvec, centre, x0, z1, z2)
            return (xs.clone() if xs is not None else None), res.clone()
        if sampler == "ode":
            T0 = self.T if T0 is None else T0
            pr = self._prior_to_device((R, 9), T=T0)
            x0 = pr if init_x is None else init_x.float() + pr
            coupling = getattr(self, "coupling_group", None)  # batch sharded over the ranks of that group: error norm over ALL its rows
            key = ("ode", B, K, id(coupling) if coupling is not None else None)
            smp = self._samplers.get(key)
            if smp is None:
                smp = self._samplers[key] = ODESampler(self.pose_score_net, B, K, self.device, coupling_group=coupling)
            self.last_sampler = smp
            return smp.run(cvec, centre, x0, T0, num_steps=self.cfg.sampling_steps, eps=self.sampling_eps, return_process=return_process)
        raise NotImplementedError(sampler)

    def _sample_energy_model(self, cvec, K, centre, sampler, init_x, T0, noise, return_process):
        """Sampling from the ENERGY model: the score is the autograd gradient of the inner-product energy (posenet.py:94-130
        with self = PoseEnergyNet, energynet.py:200-222) - never f/sigma.  The same device-resident samplers as the score model's
        (one captured launch chain for PC, the RK45 driver for the ODE), with the forward pass + vector-Jacobian product of the energy
        network inside the step / stage kernels (csrc/score_bwd.h)."""
        R = cvec.shape[0] * K
        if sampler == "pc":
            n = self.cfg.sampling_steps
            if n is None:
                raise ValueError("the PC sampler needs cfg.sampling_steps")
            x0 = self._prior_to_device((R, 9)) if init_x is None else init_x.float()
            B = cvec.shape[0]
            key = ("pc-energy", B, K, n, return_process)
            smp = self._samplers.get(key)
            if smp is None:  # the same captured launch chain as the score model's, with the energy model's score inside the step kernel
                smp = self._samplers[key] = PCSampler(self.pose_score_net, B, K, n, self.device, record_traj=return_process, model="energy")
            z1, z2 = noise if noise is not None else (None, None)
            xs, res = smp.run(cvec, centre, x0, z1, z2)
            return (xs.clone() if xs is not None else None), res.clone()
        if sampler == "ode":
            T0 = self.T if T0 is None else T0
            pr = self._prior_to_device((R, 9), T=T0)
            x0 = pr if init_x is None else init_x.float() + pr
            B = cvec.shape[0]
            key = ("ode-energy", B, K)
            smp = self._samplers.get(key)
            if smp is None:  # the device-resident RK45 driver with the energy model's score inside the stage kernels
                smp = self._samplers[key] = ODESampler(self.pose_score_net, B, K, self.device, model="energy")
            out = smp.run(cvec, centre, x0, T0, num_steps=self.cfg.sampling_steps, eps=self.sampling_eps, return_process=return_process)
            self.last_energy_ode_stats = {"nfev": int(smp.last_stats["nfev"]), "attempts": int(smp.last_stats["n_attempts"])}
            return out
        raise NotImplementedError(sampler)

    def calc_likelihood(self, data, atol=1e-5, rtol=1e-5):
        """posenet.py:133-147: log-likelihood (bits) of data['sampled_pose'] under the score model, one probe per row drawn from
        the prior.  data['pts_feat'] must be there (mode 'pts_feature')."""
        from .likelihood import cond_ode_likelihood
        self._need_weights()
        if self.cfg.posenet_mode != "score":
            raise NotImplementedError("likelihoods come from the score model")
        cvec, K = self._rows(data)
        x = data["sampled_pose"].float().contiguous()
        epsilon = self.prior_fn((x.shape[0], 9)).to(self.device)
        self.last_likelihood_stats = {}
        key = ("likelihood", cvec.shape[0], K)
        solver = self._samplers.get(key)
        if solver is None:
            solver = self._samplers[key] = ODESampler(self.pose_score_net, cvec.shape[0], K, self.device, model="likelihood")
        _, ll = cond_ode_likelihood(self.pose_score_net, cvec, K, x, epsilon, eps=self.sampling_eps, rtol=rtol, atol=atol,
                                    stats=self.last_likelihood_stats, solver=solver)
        return ll

    @staticmethod
    def _uniform_t(t, trusted=False):
        """The hoisted time embedding serves ONE diffusion time per launch: refuse per-row times instead of silently using row 0
        (they only occur in training and in get_energy(T=None), which groups the rows by time itself)."""
        tt = t.reshape(-1)
        if trusted or (tt.is_cuda and torch.cuda.is_current_stream_capturing()):  # the check reads the device: not inside a capture
            return tt[:1].float().contiguous()
        if tt.numel() > 1 and not bool((tt == tt[0]).all()):
            raise NotImplementedError("per-row diffusion times: the HIP path evaluates one time value per launch (uniform t)")
        return tt[:1].float().contiguous()

    # ------------------------------------------------------------------ string dispatch (posenet.py:150-179)
    def forward(self, data, mode="score", init_x=None, T0=None):
        if mode == "pts_feature":
            return self.extract_pts_feature(data)
        if mode in ("score", "energy"):
            self._need_weights()
            if mode == "energy" and self.cfg.posenet_mode != "energy":
                raise NotImplementedError("an energy from the score model does not exist in the reference either (posenet.py:154-160)")
            K = int(data.get("_repeat", 1))
            if mode == "score" and self.cfg.posenet_mode == "energy":
                # score of the energy model: gradient of the inner-product energy w.r.t. the pose (energynet.py:200-222)
                cvec = self.pose_score_net.cloud_embed(data["pts_feat"].float())
                t0 = self._uniform_t(data["t"], bool(data.get("_t_uniform", False)))
                tvec = self.pose_score_net.time_embed(t0)
                sigma = (SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** t0).contiguous()
                return self.pose_score_net.energy_score(cvec, K, data["sampled_pose"].float().contiguous(), tvec[0], sigma)
            if K == 1:
                return self.pose_score_net.forward_rows(data["pts_feat"], data["sampled_pose"], data["t"], mode)
            cvec = self.pose_score_net.cloud_embed(data["pts_feat"].float())
            t0 = self._uniform_t(data["t"], bool(data.get("_t_uniform", False)))
            tvec = self.pose_score_net.time_embed(t0)
            sigma = (SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** t0).contiguous()
            return self.pose_score_net.evaluate(cvec, K, data["sampled_pose"].float().contiguous(), tvec[0], sigma, mode)
        if mode == "likelihood":
            return self.calc_likelihood(data)
        if mode == "pc_sample":
            return self.sample(data, "pc", init_x=init_x)
        if mode == "ode_sample":
            return self.sample(data, "ode", init_x=init_x, T0=T0)
        raise NotImplementedError(mode)

    __call__ = forward
