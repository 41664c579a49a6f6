"""PoseScoreNet / PoseEnergyNet on the HIP kernels (reference: networks/gf_algorithms/scorenet.py:85-222,
energynet.py:34-222; regression_head 'Rx_Ry_and_T', pose_mode 'rot_matrix', per_point_feature False).

The per-cloud and per-time parts of the first head layer are hoisted (exact algebra, see csrc/scorenet.hip):
    cvec = cloud_embed(pts_feat)   once per cloud
    tvec = time_embed(t)           once per time value (t is identical for every row of an evaluation)
"""
import torch

from . import _lib
from ._lib import ptr, stream_ptr
from .sde import SIGMA_MAX, SIGMA_MIN
from .weights import ScoreNetWeights


class ScoreNetHIP:
    def __init__(self, state_dict, device="cuda", prefix="pose_score_net."):
        self.device = torch.device(device)
        self.w = ScoreNetWeights(state_dict, self.device, prefix)

    def cloud_embed(self, pts_feat):
        """pts_feat [B,1024] -> cvec [B,768]"""
        _lib.check_device()
        pts_feat = pts_feat.contiguous()
        B = pts_feat.shape[0]
        cvec = torch.empty(B, 768, device=self.device)
        _lib.call("gp_cloud_embed", B, self.w.ref(), ptr(pts_feat), ptr(cvec), stream_ptr())
        return cvec

    def time_embed(self, t_dev, out=None):
        """t_dev [nt] f32 device tensor -> tvec [nt,768]"""
        _lib.check_device()
        t_dev = t_dev.contiguous()
        nt = t_dev.numel()
        tvec = torch.empty(nt, 768, device=self.device) if out is None else out
        _lib.call("gp_time_embed", nt, self.w.ref(), ptr(t_dev), ptr(tvec), stream_ptr())
        return tvec

    def evaluate(self, cvec, k, x, tvec, sigma_dev, mode="score", out=None, tile=0):
        """x [B*k,9] f32; cvec [B,768]; tvec [768]; sigma_dev [1] f32 -> score [B*k,9] or energy [B*k,2].
        tile: launch plan (0 = automatic; 16 / 32 tile form, 128 chain form - include/genpose_hip.h: gp_score_eval_plan)"""
        _lib.check_device()
        B = cvec.shape[0]
        R = B * k
        m = 0 if mode == "score" else 1
        if out is None:
            out = torch.empty(R, 9 if m == 0 else 2, device=self.device)
        _lib.call("gp_score_eval_plan", int(tile), B, k, self.w.ref(), ptr(cvec), ptr(tvec), ptr(x), ptr(sigma_dev), m, ptr(out), stream_ptr())
        return out

    def score_and_divergence(self, cvec, k, x, eps, tvec, sigma_dev):
        """score [B*k,9] and the Hutchinson divergence estimate eps^T (d score / d x) eps [B*k] in one launch (gp_score_div)."""
        _lib.check_device()
        B = cvec.shape[0]
        R = B * k
        score = torch.empty(R, 9, device=self.device)
        div = torch.empty(R, device=self.device)
        _lib.call("gp_score_div", B, k, self.w.ref(), ptr(cvec), ptr(tvec), ptr(x), ptr(eps), ptr(sigma_dev), ptr(score), ptr(div), stream_ptr())
        return score, div

    def energy_score(self, cvec, k, x, tvec, sigma_dev, with_energy=False):
        """Score of the energy model = d/dx <x, f(x)/sigma> [B*k,9] (+ that energy [B*k]) in one launch (gp_energy_score)."""
        _lib.check_device()
        R = cvec.shape[0] * k
        score = torch.empty(R, 9, device=self.device)
        energy = torch.empty(R, device=self.device) if with_energy else None
        _lib.call("gp_energy_score", cvec.shape[0], k, self.w.ref(), ptr(cvec), ptr(tvec), ptr(x), ptr(sigma_dev), ptr(score), ptr(energy), stream_ptr())
        return (score, energy) if with_energy else score

    def forward_rows(self, pts_feat_rows, pose, t, mode="score"):
        """Reference-shaped call: pts_feat [R,1024] (one feature row per pose row), pose [R,9], t [R,1] (uniform)."""
        tt = t.reshape(-1)
        t0 = tt[:1].contiguous().float()
        if tt.numel() > 1 and not bool((tt == tt[0]).all()):
            raise NotImplementedError("per-row diffusion times are a training-only feature; the HIP path needs a uniform t")
        cvec = self.cloud_embed(pts_feat_rows.float())
        tvec = self.time_embed(t0)
        sigma = (SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** t0).contiguous()
        return self.evaluate(cvec, 1, pose.float().contiguous(), tvec[0], sigma, mode)
