"""Training steps of the score and the energy model: denoising-score-matching loss, the energy model's pairwise
ranking loss, optimiser step, learning-rate schedule, weight average, checkpoints - what `PoseNet.train_func(data, pose_samples,
gf_mode)` does in the reference (networks/posenet_agent.py:176-317, 530-550; networks/gf_algorithms/losses.py:47-89;
networks/reward.py:63-128; utils/metrics.py:83-186).

What runs where.  The PointNet++ grouping operators - furthest point sampling, gather, ball query, group, and the BACKWARD of gather
and group - are the hand-written gfx950 kernels of libgenpose_hip.so, reached through `genpose_amd.pointnet2_cuda` (the drop-in for the
reference's CUDA extension) and wrapped as autograd Functions below, exactly where the reference wraps `pointnet2_cuda`
(pointnet2_utils.py:11-265).  The dense layers (1x1 convolutions + BatchNorm in TRAINING mode, the score MLP) and their backward
run on torch autograd: the fused inference kernels fold BatchNorm with its running statistics and keep no activations, neither of
which a training step can use.  Training is the row of SURVEY §8f farthest from the benchmarked path and is not timed.

The modules carry the reference's parameter names, so `Trainer.state_dict()` loads straight into the inference agent
(`PoseNet.load_state_dict`) and a reference checkpoint loads into the trainer.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_cuda as pn2
from .sde import EPS, init_sde

# networks/pts_encoder/pointnet2.py:57-66 (ClsMSG_CFG_Light)
LIGHT = dict(npoints=[512, 256, 128, None], radii=[[0.02, 0.04], [0.04, 0.08], [0.08, 0.16], [None, None]],
             nsamples=[[16, 32], [16, 32], [16, 32], [None, None]],
             mlps=[[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]], [[128, 196, 256], [128, 196, 256]],
                   [[256, 256, 512], [256, 384, 512]]])


# ---------------------------------------------------------------------------------------------- grouping operators with autograd
class _FurthestPointSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        B, N, _ = xyz.shape
        idx = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pn2.furthest_point_sampling_wrapper(B, N, npoint, xyz.contiguous(), temp, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, g):
        return None, None


class _Gather(torch.autograd.Function):
    """features [B,C,N], idx [B,m] -> [B,C,m]"""

    @staticmethod
    def forward(ctx, features, idx):
        B, C, N = features.shape
        m = idx.shape[1]
        out = torch.empty(B, C, m, device=features.device)
        pn2.gather_points_wrapper(B, C, N, m, features.contiguous(), idx, out)
        ctx.save_for_backward(idx)
        ctx.dims = (C, N)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        C, N = ctx.dims
        B, m = idx.shape
        grad = torch.zeros(B, C, N, device=g.device)
        pn2.gather_points_grad_wrapper(B, C, N, m, g.contiguous(), idx, grad)
        return grad, None


class _BallQuery(torch.autograd.Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        B, N, _ = xyz.shape
        m = new_xyz.shape[1]
        idx = torch.zeros(B, m, nsample, dtype=torch.int32, device=xyz.device)
        pn2.ball_query_wrapper(B, N, m, radius, nsample, new_xyz.contiguous(), xyz.contiguous(), idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, g):
        return None, None, None, None


class _Group(torch.autograd.Function):
    """features [B,C,N], idx [B,m,ns] -> [B,C,m,ns]"""

    @staticmethod
    def forward(ctx, features, idx):
        B, C, N = features.shape
        _, m, ns = idx.shape
        out = torch.empty(B, C, m, ns, device=features.device)
        pn2.group_points_wrapper(B, C, N, m, ns, features.contiguous(), idx, out)
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, C, m, ns = g.shape
        grad = torch.zeros(B, C, ctx.N, device=g.device)
        pn2.group_points_grad_wrapper(B, C, ctx.N, m, ns, g.contiguous(), idx, grad)
        return grad, None


# ---------------------------------------------------------------------------------------------- modules (reference parameter names)
class _BN(nn.Module):  # pytorch_utils.py: BatchNorm2d wrapper -> keys '...bn.bn.weight'
    def __init__(self, c):
        super().__init__()
        self.bn = nn.BatchNorm2d(c)

    def forward(self, x):
        return self.bn(x)


class _ConvBNReLU(nn.Module):  # keys 'layer{l}.conv.weight', 'layer{l}.bn.bn.*'
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=1, bias=False)
        self.bn = _BN(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class _SharedMLP(nn.Sequential):
    def __init__(self, spec):
        super().__init__()
        for l in range(len(spec) - 1):
            self.add_module(f"layer{l}", _ConvBNReLU(spec[l], spec[l + 1]))


class _SAModule(nn.Module):
    """One set-abstraction level with multi-scale grouping (pointnet2_modules.py:19-116) or GroupAll when npoint is None."""

    def __init__(self, npoint, radii, nsamples, specs):
        super().__init__()
        self.npoint, self.radii, self.nsamples = npoint, radii, nsamples
        self.mlps = nn.ModuleList([_SharedMLP(s) for s in specs])

    def forward(self, xyz, features):
        """xyz [B,N,3]; features [B,C,N] or None -> (new_xyz, [B, sum C_out, npoint])"""
        xyz_t = xyz.transpose(1, 2).contiguous()
        new_xyz = None
        if self.npoint is not None:
            new_xyz = _Gather.apply(xyz_t, _FurthestPointSample.apply(xyz, self.npoint)).transpose(1, 2).contiguous()
        outs = []
        for i, mlp in enumerate(self.mlps):
            if self.npoint is not None:
                idx = _BallQuery.apply(self.radii[i], self.nsamples[i], xyz, new_xyz)
                g = _Group.apply(xyz_t, idx) - new_xyz.transpose(1, 2).unsqueeze(-1)      # QueryAndGroup, pointnet2_utils.py:246-258
                x = g if features is None else torch.cat([g, _Group.apply(features, idx)], dim=1)
            else:
                g = xyz_t.unsqueeze(2)                                                    # GroupAll, pointnet2_utils.py:276-289
                x = g if features is None else torch.cat([g, features.unsqueeze(2)], dim=1)
            x = mlp(x)
            outs.append(F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)


class TrainableEncoder(nn.Module):
    """Pointnet2ClsMSG(0) (networks/pts_encoder/pointnet2.py:166-211), 'light' configuration."""

    def __init__(self, cfg=LIGHT):
        super().__init__()
        self.SA_modules = nn.ModuleList()
        cin = 0
        for k in range(len(cfg["npoints"])):
            specs = [[cin + 3] + list(m) for m in cfg["mlps"][k]]
            self.SA_modules.append(_SAModule(cfg["npoints"][k], cfg["radii"][k], cfg["nsamples"][k], specs))
            cin = sum(m[-1] for m in cfg["mlps"][k])

    def forward(self, pts):
        xyz, feats = pts[..., 0:3].contiguous(), None
        for sa in self.SA_modules:
            new_xyz, feats = sa(xyz, feats)
            if new_xyz is not None:
                xyz = new_xyz
        return feats.squeeze(-1)


class _Fourier(nn.Module):  # scorenet.py:55-64
    def __init__(self, embed_dim=128, scale=30.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embed_dim // 2) * scale, requires_grad=False)

    def forward(self, x):
        p = x[:, None] * self.W[None, :] * 2 * math.pi
        return torch.cat([torch.sin(p), torch.cos(p)], dim=-1)


class TrainableScoreNet(nn.Module):
    """PoseScoreNet, regression_head 'Rx_Ry_and_T', pose_mode 'rot_matrix' (scorenet.py:85-222)."""

    def __init__(self, marginal_prob_fn):
        super().__init__()
        self.marginal_prob_fn = marginal_prob_fn
        self.pose_encoder = nn.Sequential(nn.Linear(9, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU())
        self.t_encoder = nn.Sequential(_Fourier(128), nn.Linear(128, 128), nn.ReLU())
        for h in ("rot_x", "rot_y", "trans"):
            tail = nn.Sequential(nn.Linear(128 + 256 + 1024, 256), nn.ReLU(), nn.Linear(256, 3))
            nn.init.zeros_(tail[2].weight)  # zero_module (scorenet.py:156-170)
            nn.init.zeros_(tail[2].bias)
            setattr(self, f"fusion_tail_{h}", tail)

    def forward(self, data):
        t = data["t"]
        total = torch.cat([data["pts_feat"], self.t_encoder(t.squeeze(1)), self.pose_encoder(data["sampled_pose"])], dim=-1)
        _, std = self.marginal_prob_fn(total, t)
        out = torch.cat([self.fusion_tail_rot_x(total), self.fusion_tail_rot_y(total), self.fusion_tail_trans(total)], dim=-1)
        return out / (std + 1e-7)


class TrainableEnergyNet(TrainableScoreNet):
    """PoseEnergyNet (energynet.py:34-222) with the shipped options: energy_mode 'IP', s_theta_mode 'score', norm_energy 'identical'.
    Same parameters (and parameter names) as the score net; what it RETURNS differs:
      return_item 'energy'  [R,2]  decoupled inner-product energies <x_rot, s_rot>, <x_trans, s_trans>, s = f_theta / std   (:143-198)
      return_item 'score'   [R,9]  d/dx <x, f_theta(x) / std> by autograd, kept in the graph (create_graph) so that the
                                   score-matching loss can be back-propagated through it                                   (:200-222)"""

    def f_over_std(self, pts_feat, pose, t):
        total = torch.cat([pts_feat, self.t_encoder(t.squeeze(1)), self.pose_encoder(pose)], dim=-1)
        _, std = self.marginal_prob_fn(total, t)
        f = torch.cat([self.fusion_tail_rot_x(total), self.fusion_tail_rot_y(total), self.fusion_tail_trans(total)], dim=-1)
        return f / std

    def forward(self, data, return_item="score"):
        pts_feat, pose, t = data["pts_feat"], data["sampled_pose"], data["t"]
        if return_item == "energy":
            s = self.f_over_std(pts_feat, pose, t)
            return torch.stack([(pose[:, :-3] * s[:, :-3]).sum(-1), (pose[:, -3:] * s[:, -3:]).sum(-1)], dim=-1)
        if return_item != "score":
            raise NotImplementedError(return_item)
        with torch.enable_grad():
            x = pose.detach().requires_grad_(True)
            energy = (x * self.f_over_std(pts_feat, x, t)).sum(-1)
            (score,) = torch.autograd.grad(energy, x, grad_outputs=torch.ones_like(energy), create_graph=True)
        return score


class TrainableGFObjectPose(nn.Module):
    def __init__(self, marginal_prob_fn, posenet_mode="score"):
        super().__init__()
        if posenet_mode not in ("score", "energy"):
            raise NotImplementedError(posenet_mode)
        self.posenet_mode = posenet_mode
        self.pts_encoder = TrainableEncoder()
        self.pose_score_net = (TrainableScoreNet if posenet_mode == "score" else TrainableEnergyNet)(marginal_prob_fn)

    def forward(self, data, mode="score"):
        if mode == "pts_feature":
            return self.pts_encoder(data["pts"])
        if mode == "score":
            return self.pose_score_net(data)
        if mode == "energy":
            if self.posenet_mode != "energy":
                raise NotImplementedError("an energy from the score model does not exist in the reference either (posenet.py:154-160)")
            return self.pose_score_net(data, return_item="energy")
        raise NotImplementedError(mode)


# ---------------------------------------------------------------------------------------------- losses, weight average, trainer
def dsm_loss(model, data, marginal_prob_fn, eps=EPS, draws=None):
    """Denoising score matching, losses.py:47-89: t ~ U(eps, 1), x = mu + z std, loss = mean_b sum_d std^2 (s(x,t) + z/std)^2.
    draws (tests): (u [bs] uniform(0,1), z [bs,9] standard normal) instead of the generator."""
    gt = data["zero_mean_gt_pose"]
    bs = gt.shape[0]
    u = torch.rand(bs, device=gt.device) if draws is None else draws[0].to(gt.device)
    t = (u * (1.0 - eps) + eps).unsqueeze(-1)
    mu, std = marginal_prob_fn(gt, t)
    std = std.view(-1, 1)
    z = torch.randn_like(gt) if draws is None else draws[1].to(gt.device)
    data["sampled_pose"] = mu + z * std
    data["t"] = t
    est = model(data)
    target = -z * std / (std ** 2)
    return torch.mean(torch.sum(((std ** 2) * (est - target) ** 2).view(bs, -1), dim=-1))


SYMMETRIC = ("bottle", "can", "bowl")  # + a mug whose handle is not visible (utils/metrics.py:107-113)


def pose_errors(pred9, gt9, class_ids, handle_visibility, synset_names):
    """get_metrics (utils/metrics.py:157-186, object-to-camera poses) -> compute_RT_errors (:76-118): rotation error in degrees - about
    the y axis only for the symmetric categories - and translation error in centimetres, per row.  The reference hands float32 4x4
    matrices to numpy, so the general branch (R1 R2^T, trace, arccos) and the shift are float32 arithmetic and only the symmetric branch
    (R @ int64 y-axis) is promoted to float64; the same dtypes here, so that near-ties rank as they do there."""
    from . import rotation
    R1 = rotation.get_rot_matrix(pred9[:, :6].float()).float().cpu().numpy()
    R2 = rotation.get_rot_matrix(gt9[:, :6].float()).float().cpu().numpy()
    T1, T2 = pred9[:, 6:].float().cpu().numpy(), gt9[:, 6:].float().cpu().numpy()
    R1 = R1 / np.cbrt(np.linalg.det(R1))[:, None, None]  # float32 (metrics.py:100-103)
    R2 = R2 / np.cbrt(np.linalg.det(R2))[:, None, None]
    ids = np.asarray(class_ids).reshape(-1).astype(np.int64)
    vis = np.asarray(handle_visibility).reshape(-1)
    names = np.asarray(synset_names)[ids]
    sym = np.isin(names, SYMMETRIC) | ((names == "mug") & (vis == 0))
    y1, y2 = R1[:, :, 1].astype(np.float64), R2[:, :, 1].astype(np.float64)  # R @ np.array([0, 1, 0]): float32 @ int64 -> float64
    cos_sym = (y1 * y2).sum(-1) / (np.linalg.norm(y1, axis=-1) * np.linalg.norm(y2, axis=-1))
    cos_full = (np.trace(np.matmul(R1, R2.transpose(0, 2, 1)), axis1=1, axis2=2) - np.float32(1)) / np.float32(2)
    theta_sym = np.arccos(np.clip(cos_sym, -1.0, 1.0)) * 180 / np.pi
    theta_full = (np.arccos(np.clip(cos_full, np.float32(-1.0), np.float32(1.0))) * np.float32(180) / np.float32(np.pi)).astype(np.float64)
    theta = np.where(sym, theta_sym, theta_full)
    shift = (np.linalg.norm(T1 - T2, axis=-1) * np.float32(100)).astype(np.float64)
    return theta, shift


def ranking_loss(energy_by_error):
    """networks/reward.py:109-128 on energies sorted by pose error (low error first), [bs, K, 2]: over all pairs i < j,
    1 + (E_j - E_i) / (|E_i - E_j| + 1e-5) - zero when the candidate with the smaller error has the larger energy.  All K(K-1)/2 pairs
    at once instead of the reference's Python double loop (the mean over pairs of per-pair means over [bs, 2] is the same sum)."""
    K = energy_by_error.shape[1]
    i, j = torch.triu_indices(K, K, offset=1, device=energy_by_error.device)
    ei, ej = energy_by_error[:, i, :], energy_by_error[:, j, :]
    return (1 + (ej - ei) / (torch.abs(ei - ej) + 1e-5)).mean(dim=(0, 2)).sum() / i.numel()


class WeightAverage:
    """Polyak average of the trainable parameters with the warm-up of the reference (decay = min(rate, (1 + n) / (10 + n)),
    score_utils.py:35-49), kept as one list of tensors updated by a single fused lerp.  `applied()` runs a block with the averaged
    weights swapped INTO the network (evaluation, checkpoints) and swaps the raw weights back afterwards - a pointer exchange, no
    copies."""

    def __init__(self, params, rate):
        if not 0.0 <= rate <= 1.0:
            raise ValueError(f"average rate {rate} outside [0, 1]")
        self.rate, self.steps = rate, 0
        self.params = [p for p in params if p.requires_grad]
        self.average = [p.detach().clone() for p in self.params]

    def step(self):
        self.steps += 1
        keep = min(self.rate, (1 + self.steps) / (10 + self.steps))
        with torch.no_grad():
            torch._foreach_lerp_(self.average, [p.detach() for p in self.params], 1.0 - keep)

    def _swap(self):
        for p, a in zip(self.params, self.average):
            p.data, a.data = a.data, p.data

    def applied(self):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            self._swap()
            try:
                yield
            finally:
                self._swap()
        return ctx()


class Trainer:
    """The training half of the reference agent (networks/posenet_agent.py:46-317, 530-550): Adam (or SGD), linear warm-up then
    exponential lr decay, gradient clipping, weight average; score model, energy model without (`energy_wo_ranking`) or with the
    ranking loss (`energy`)."""

    def __init__(self, device="cuda", lr=1e-3, optimizer="Adam", lr_decay=0.98, grad_clip=1.0, ema_rate=0.999, repeat_num=20, sde_mode="ve",
                 posenet_mode="score", warmup=100, synset_names=("bottle", "bowl", "camera", "can", "laptop", "mug")):
        self.device = torch.device(device)
        self.prior_fn, self.marginal_prob_fn, self.sde_fn, self.sampling_eps, self.T = init_sde(sde_mode)
        self.net = TrainableGFObjectPose(self.marginal_prob_fn, posenet_mode).to(self.device)
        if optimizer == "Adam":
            self.optimizer = torch.optim.Adam(self.net.parameters(), betas=(0.9, 0.999), eps=1e-8, lr=lr)
        elif optimizer == "SGD":
            self.optimizer = torch.optim.SGD(self.net.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
        else:
            raise NotImplementedError(optimizer)
        self.base_lr, self.warmup = lr, warmup
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, lr_decay)
        self.ema = WeightAverage(self.net.parameters(), ema_rate)
        self.grad_clip, self.repeat_num, self.ema_rate = grad_clip, repeat_num, ema_rate
        self.synset_names = list(synset_names)
        self.clock = {"epoch": 1, "minibatch": 0, "step": 0}  # TrainClock (utils/genpose_utils.py:70-96)

    # ------------------------------------------------------------------ state
    def load_state_dict(self, sd, reset_ema=True):
        self.net.load_state_dict({k: v.to(self.device) for k, v in sd.items()})
        if reset_ema:
            self.ema = WeightAverage(self.net.parameters(), self.ema_rate)

    def state_dict(self, ema=True):
        """Reference-layout state dict; with ema=True the averaged weights, as save_ckpt stores them (posenet_agent.py:125-140)."""
        if not ema:
            return {k: v.detach().cpu().clone() for k, v in self.net.state_dict().items()}
        with self.ema.applied():
            return {k: v.detach().cpu().clone() for k, v in self.net.state_dict().items()}

    def save_ckpt(self, path):
        """The reference's checkpoint dictionary (posenet_agent.py:117-141): averaged weights + optimiser + scheduler + clock."""
        torch.save({"clock": dict(self.clock), "model_state_dict": self.state_dict(ema=True), "optimizer_state_dict": self.optimizer.state_dict(),
                    "scheduler_state_dict": self.scheduler.state_dict()}, path)

    def load_ckpt(self, path, load_model_only=False):
        """posenet_agent.py:143-173 (`module.`-prefixed keys of a DataParallel checkpoint are accepted)."""
        import os
        if not os.path.exists(path):
            raise ValueError("Checkpoint {} not exists.".format(path))
        ck = torch.load(path, map_location="cpu")
        self.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in ck["model_state_dict"].items()})
        if not load_model_only:
            self.optimizer.load_state_dict(ck["optimizer_state_dict"])
            self.scheduler.load_state_dict(ck["scheduler_state_dict"])
            self.clock = dict(ck["clock"])

    def tick(self):
        self.clock["minibatch"] += 1
        self.clock["step"] += 1

    def tock(self):
        """End of an epoch (TrainClock.tock, utils/genpose_utils.py:82-84)."""
        self.clock["epoch"] += 1
        self.clock["minibatch"] = 0

    def update_learning_rate(self):
        """posenet_agent.py:543-550: linear warm-up over `warmup` steps, then exponential decay for as long as lr >= 1e-4."""
        group = self.optimizer.param_groups[-1]
        if self.clock["step"] <= self.warmup:
            group["lr"] = self.base_lr / self.warmup * self.clock["step"]
        elif not group["lr"] < 1e-4:
            # WHEN the decay is applied belongs to the caller: the reference's loop calls this at the end of an epoch, after that epoch's
            # optimizer steps (runners/trainer.py:296-303) - scheduler.step() then follows optimizer.step() as torch expects
            self.scheduler.step()

    # ------------------------------------------------------------------ losses
    def collect_score_loss(self, data, draws=None):
        loss = 0
        for r in range(self.repeat_num):
            loss = loss + dsm_loss(self.net, data, self.marginal_prob_fn, draws=None if draws is None else (draws[0][r], draws[1][r]))
        return {"gf": loss / self.repeat_num}

    def get_energy(self, data, pose_samples, T=None, t_draws=None):
        """PoseNet.get_energy(mode='train', extract_pts_feature=False) (posenet_agent.py:471-527): energies [bs, K, 2] of the
        candidates, differentiable w.r.t. the network; T=None draws one T in {1e-5 .. 9e-5} per cloud (t_draws [bs,1] ints 1..9: tests)."""
        bs, K = pose_samples.shape[:2]
        feat = data["pts_feat"].unsqueeze(1).expand(bs, K, -1).reshape(bs * K, -1)
        pose = pose_samples.clone().reshape(bs * K, -1).type_as(feat)
        if T is not None:
            t = torch.ones(bs * K, 1).type_as(feat) * T
        else:
            ti = torch.randint(1, 10, (bs, 1)) if t_draws is None else t_draws
            t = (ti.type_as(feat) / 1e5).repeat(1, K).view(bs * K, 1)
        pose[:, -3:] -= data["pts_center"].unsqueeze(1).expand(bs, K, -1).reshape(bs * K, -1)
        return self.net({"pts_feat": feat, "sampled_pose": pose, "t": t}, mode="energy").reshape(bs, K, -1)

    def collect_ranking_loss(self, data, pred_pose, t_draws=None):
        """posenet_agent.py:227-259: energies of the candidates ordered by their pose error (per channel) -> pairwise ranking loss."""
        energy = self.get_energy(data, pred_pose, t_draws=t_draws)
        bs, K = pred_pose.shape[:2]
        rep = lambda v: v.reshape(bs, -1).unsqueeze(1).expand(bs, K, -1).reshape(bs * K, -1)
        rot_err, trans_err = pose_errors(pred_pose.reshape(bs * K, -1), rep(data["gt_pose"]), rep(data["id"]).cpu().numpy(),
                                         rep(data["handle_visibility"]).cpu().numpy(), self.synset_names)
        metrics = torch.from_numpy(np.stack([rot_err, trans_err], axis=-1)).to(energy.device).reshape(bs, K, 2)
        order = torch.argsort(metrics, dim=1, descending=False)  # reward.py:63-83 (sort_results)
        return {"ranking": ranking_loss(energy.gather(1, order))}

    def _update_network(self, losses):
        loss = sum(losses.values())
        self.optimizer.zero_grad()
        loss.backward()
        if self.grad_clip >= 0:
            torch.nn.utils.clip_grad_norm_(self.net.parameters(), max_norm=self.grad_clip)
        self.optimizer.step()

    def train_func(self, data, pose_samples=None, gf_mode="score", draws=None, t_draws=None):
        """One step (posenet_agent.py:310-317).  data on the device: 'pts' [B,1024,3], 'zero_mean_pts', 'zero_mean_gt_pose' [B,9]; for
        gf_mode 'energy' also 'pts_center', 'gt_pose', 'id', 'handle_visibility' and pose_samples [B,K,9] (candidates of the score
        model).  'score' / 'energy_wo_ranking' = train_score_func on whatever network the trainer was built with (posenet_mode
        'score' -> PoseScoreNet; 'energy' -> PoseEnergyNet, whose score is the autograd gradient of its energy)."""
        if gf_mode not in ("score", "energy_wo_ranking", "energy"):
            raise NotImplementedError(gf_mode)
        want = "score" if gf_mode == "score" else "energy"
        if self.net.posenet_mode != want:
            raise ValueError(f"gf_mode '{gf_mode}' trains a posenet_mode='{want}' network; this trainer holds a '{self.net.posenet_mode}' one")
        self.net.train()
        data["pts_feat"] = self.net(data, mode="pts_feature")
        losses = self.collect_score_loss(data, draws)
        if gf_mode == "energy":
            if pose_samples is None:
                raise ValueError("gf_mode 'energy' needs pose_samples [B,K,9]")
            losses.update(self.collect_ranking_loss(data, pose_samples, t_draws))
        self._update_network(losses)
        self.ema.step()
        return losses
