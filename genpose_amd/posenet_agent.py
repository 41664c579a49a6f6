"""PoseNet agent on the HIP kernels: the call surface runners/evaluation_single.py and runners/evaluation_tracking.py
use (reference: networks/posenet_agent.py:46-527) -

    agent = PoseNet(cfg)                                   # cfg: the reference's argparse namespace (configs/config.py)
    agent.load_ckpt(model_dir=path, model_path=True, load_model_only=True)
    pred_pose = agent.pred_func(data, repeat_num, save_path, return_average_res, init_x, T0, return_process)
    energy    = agent.get_energy(data, pose_samples, T, mode, extract_pts_feature)
    agent.net(data, mode=..., init_x=..., T0=...)

Training / logging / visualisation methods of the reference agent are out of scope.
"""
import os

import torch

from . import rotation
from .posenet import GFObjectPose
from .sde import SIGMA_MAX, SIGMA_MIN, init_sde


class PoseNet:
    def __init__(self, cfg):
        self.cfg = cfg
        self.is_testing = False
        self.pts_feature = False
        if getattr(cfg, "is_train", False):
            raise NotImplementedError("genpose_amd implements the inference hot path only (is_train must be False)")
        self.model_dir = f"./results/ckpts/{getattr(cfg, 'log_dir', 'debug')}"
        self.prior_fn, self.marginal_prob_fn, self.sde_fn, self.sampling_eps, self.T = init_sde(cfg.sde_mode)
        self.net = self.build_net()

    def get_network(self, name):
        if name == "GFObjectPose":
            return GFObjectPose(self.cfg, self.prior_fn, self.marginal_prob_fn, self.sde_fn, self.sampling_eps, self.T)
        raise NotImplementedError(f"Got name '{name}'")

    def build_net(self):
        return self.get_network("GFObjectPose")

    # ------------------------------------------------------------------ checkpoints (posenet_agent.py:143-173)
    def load_ckpt(self, name=None, model_dir=None, model_path=False, load_model_only=False):
        if not model_path:
            if name not in ("latest", "best"):
                name = "ckpt_epoch{}".format(name)
            load_path = os.path.join(self.model_dir if model_dir is None else model_dir, "{}.pth".format(name))
        else:
            load_path = model_dir
        if not os.path.exists(load_path):
            raise ValueError("Checkpoint {} not exists.".format(load_path))
        checkpoint = torch.load(load_path, map_location="cpu")
        print("Loading checkpoint from {} ...".format(load_path))
        self.net.load_state_dict(checkpoint["model_state_dict"])
        # optimizer / scheduler / clock entries are training state: ignored even when load_model_only is False

    def load_state_dict(self, state_dict):
        self.net.load_state_dict(state_dict)

    # ------------------------------------------------------------------ inference (posenet_agent.py:416-468)
    def pred_func(self, data, repeat_num, save_path="./visualization_results", return_average_res=False, init_x=None, T0=None,
                  return_process=False, noise=None):
        """-> pred_pose [B,K,9] (float64 for the ODE sampler, float32 for PC - as the reference), or the tuples of
        posenet_agent.py:460-466.  `data` gains 'pts_feat' (as in the reference, :422).  `noise` (tests only): explicit
        (z_langevin, z_predictor) draws for the PC sampler."""
        self.is_testing = True
        self.net.eval()
        if getattr(self.cfg, "save_video", False) and save_path is not None:
            raise NotImplementedError("save_video: visualisation is outside the hot path")
        with torch.no_grad():
            data["pts_feat"] = self.net(data, mode="pts_feature")
            bs = data["pts"].shape[0]
            self.pts_feature = True
            rows = {"pts_feat": data["pts_feat"], "pts_center": data["pts_center"], "_repeat": repeat_num}
            rep_init = None if init_x is None else init_x.unsqueeze(1).repeat(1, repeat_num, 1).view(bs * repeat_num, -1)
            sampler = self.cfg.sampler_mode[0]
            in_process_sample, res = self.net.sample(rows, sampler, init_x=rep_init, T0=T0, noise=noise, return_process=return_process)
            pred_pose = res.reshape(bs, repeat_num, -1)
            if in_process_sample is not None:
                in_process_sample = in_process_sample.reshape(bs, repeat_num, in_process_sample.shape[1], -1)
            self.pts_feature = False
            if return_average_res:
                rot_matrix = rotation.get_rot_matrix(res[:, :-3])
                quat_wxyz = rotation.matrix_to_quaternion(rot_matrix)
                res_q_wxyz = torch.cat((quat_wxyz, res[:, -3:]), dim=-1)
                pred_pose_q_wxyz = res_q_wxyz.reshape(bs, repeat_num, -1)
                average = torch.zeros((bs, 7), device=pred_pose_q_wxyz.device)
                average[:, :4] = rotation.average_quaternion_batch(pred_pose_q_wxyz[:, :, :4])
                average[:, 4:] = torch.mean(pred_pose_q_wxyz[:, :, 4:], dim=1)
                if return_process:
                    return pred_pose, pred_pose_q_wxyz, average, in_process_sample
                return pred_pose, pred_pose_q_wxyz, average
            if return_process:
                return [pred_pose, in_process_sample]
            return pred_pose

    # ------------------------------------------------------------------ energy (posenet_agent.py:471-527)
    def get_energy(self, data, pose_samples, T=None, mode="test", extract_pts_feature=True):
        if mode != "test":
            raise NotImplementedError("get_energy(mode='train') is training-only")
        self.is_testing = True
        self.net.eval()
        bs, repeat_num = pose_samples.shape[0], pose_samples.shape[1]
        with torch.no_grad():
            pts_feat = data["pts_feat"] if not extract_pts_feature else self.net(data, mode="pts_feature")
            self.pts_feature = True
            pose = pose_samples.clone().view(bs * repeat_num, -1).type_as(pts_feat)
            pose[:, -3:] -= data["pts_center"].unsqueeze(1).repeat(1, repeat_num, 1).view(bs * repeat_num, -1)
            rows = {"pts_feat": pts_feat, "sampled_pose": pose, "_repeat": repeat_num, "_t_uniform": True}
            if T is not None:
                rows["t"] = torch.ones(bs * repeat_num, 1).type_as(pts_feat) * T
                return self.net(rows, mode="energy").reshape(bs, repeat_num, -1)
            # posenet_agent.py:504-509: one random T per cloud from {1e-5, 2e-5, ..., 9e-5} (torch.randint(1, 10)/1e5, drawn on
            # the CPU generator).  The kernels serve one time value per launch, so the clouds are grouped by their T: one launch per
            # DISTINCT value (at most nine) over THAT value's clouds only - every row is evaluated exactly once.
            T_int = torch.randint(int(1e-5 * 1e5), int(1e-4 * 1e5), (bs, 1))
            self.last_T_samples = T_dev = T_int.type_as(pts_feat) / 1e5  # divided on the device, like the reference (.type_as comes first)
            psn = self.net.pose_score_net
            cvec = psn.cloud_embed(pts_feat.float())
            pose3 = pose.view(bs, repeat_num, -1)
            energy = torch.empty(bs, repeat_num, 2, device=pts_feat.device, dtype=torch.float32)
            for tv in torch.unique(T_int).tolist():
                idx = torch.nonzero(T_int.reshape(-1) == tv).reshape(-1).to(pts_feat.device)
                t0 = T_dev.reshape(-1)[idx[:1]].float().contiguous()  # this group's time value, as the device computed it
                sigma = (SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** t0).contiguous()
                e = psn.evaluate(cvec.index_select(0, idx).contiguous(), repeat_num, pose3.index_select(0, idx).reshape(-1, pose3.shape[-1]).contiguous(),
                                 psn.time_embed(t0)[0], sigma, "energy")
                energy.index_copy_(0, idx, e.reshape(idx.numel(), repeat_num, 2))
            return energy
