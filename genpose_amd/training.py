"""Training step of the score model: denoising-score-matching loss, optimiser step, EMA - what
`PoseNet.train_func(data, gf_mode='score')` does in the reference (networks/posenet_agent.py:285-317, 176-197, 530-540;
networks/gf_algorithms/losses.py:47-89; networks/gf_algorithms/score_utils.py:3-92).

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


class TrainableGFObjectPose(nn.Module):
    def __init__(self, marginal_prob_fn):
        super().__init__()
        self.pts_encoder = TrainableEncoder()
        self.pose_score_net = TrainableScoreNet(marginal_prob_fn)

    def forward(self, data, mode="score"):
        if mode == "pts_feature":
            return self.pts_encoder(data["pts"])
        if mode == "score":
            return self.pose_score_net(data)
        raise NotImplementedError(mode)


# ---------------------------------------------------------------------------------------------- loss, EMA, trainer
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


class ExponentialMovingAverage:
    """score_utils.py:3-92 (decay warm-up (1+n)/(10+n))."""

    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = []

    def update(self, parameters):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            for s, p in zip(self.shadow_params, [p for p in parameters if p.requires_grad]):
                s.sub_((1.0 - decay) * (s - p))

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, [p for p in parameters if p.requires_grad]):
            p.data.copy_(s.data)

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        for c, p in zip(self.collected_params, parameters):
            p.data.copy_(c.data)

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)


class Trainer:
    """The training half of the reference agent for the score model: Adam (or SGD), exponential lr decay, gradient clipping, EMA."""

    def __init__(self, device="cuda", lr=1e-3, optimizer="Adam", lr_decay=0.98, grad_clip=1.0, ema_rate=0.999, repeat_num=20, sde_mode="ve"):
        self.device = torch.device(device)
        self.prior_fn, self.marginal_prob_fn, self.sde_fn, self.sampling_eps, self.T = init_sde(sde_mode)
        self.net = TrainableGFObjectPose(self.marginal_prob_fn).to(self.device)
        if optimizer == "Adam":
            self.optimizer = torch.optim.Adam(self.net.parameters(), betas=(0.9, 0.999), eps=1e-8, lr=lr)
        elif optimizer == "SGD":
            self.optimizer = torch.optim.SGD(self.net.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
        else:
            raise NotImplementedError(optimizer)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, lr_decay)
        self.ema = ExponentialMovingAverage(self.net.parameters(), decay=ema_rate)
        self.grad_clip, self.repeat_num, self.ema_rate, self.step = grad_clip, repeat_num, ema_rate, 0

    def load_state_dict(self, sd, reset_ema=True):
        self.net.load_state_dict({k: v.to(self.device) for k, v in sd.items()})
        if reset_ema:
            self.ema = ExponentialMovingAverage(self.net.parameters(), decay=self.ema_rate)

    def state_dict(self, ema=True):
        """Reference-layout state dict; with ema=True the EMA weights, as save_ckpt stores them (posenet_agent.py:125-140)."""
        if ema:
            self.ema.store(self.net.parameters())
            self.ema.copy_to(self.net.parameters())
        sd = {k: v.detach().cpu().clone() for k, v in self.net.state_dict().items()}
        if ema:
            self.ema.restore(self.net.parameters())
        return sd

    def collect_score_loss(self, data, draws=None):
        loss = 0
        for r in range(self.repeat_num):
            loss = loss + dsm_loss(self.net, data, self.marginal_prob_fn, draws=None if draws is None else (draws[0][r], draws[1][r]))
        return {"gf": loss / self.repeat_num}

    def train_func(self, data, gf_mode="score", draws=None):
        """One step (train_score_func): data['pts'] [B,1024,3], data['zero_mean_pts'], data['zero_mean_gt_pose'] [B,9] on the device."""
        if gf_mode not in ("score", "energy_wo_ranking"):
            raise NotImplementedError("training the energy model's ranking loss is not implemented")
        self.net.train()
        data["pts_feat"] = self.net(data, mode="pts_feature")
        losses = self.collect_score_loss(data, draws)
        loss = sum(losses.values())
        self.optimizer.zero_grad()
        loss.backward()
        if self.grad_clip >= 0:
            torch.nn.utils.clip_grad_norm_(self.net.parameters(), max_norm=self.grad_clip)
        self.optimizer.step()
        self.ema.update(self.net.parameters())
        self.step += 1
        return losses
