"""Energy ranking + top-k aggregation on the device (csrc/rank.hip).
Reference: networks/reward.py:131-155 (sort_poses_by_energy), utils/sgpa_utils.py:897-954 (sort_sRT_by_energy,
'average'), runners/evaluation_tracking.py:60-77 (cal_average_sRT)."""
import torch

from . import _lib, rotation
from ._lib import ptr, stream_ptr


def rank_aggregate(poses, energy, ratio=None, selected_num=None, with_rt=False):
    """poses [B,K,9] (f32 or f64, device), energy [B,K,2] f32 ->
    dict(sorted_poses, sorted_energy, order [B,K,2] i32, avg_pose [B,7] f32 (w,x,y,z,t) or None).
    with_rt: the same launch also writes sorted_RTs [B,K,4,4] f64 (= rotation.pose9_to_RT(sorted_poses)) and, when aggregating,
    avg_RT [B,4,4] f32 (= rotation.quat_trans_to_RT(avg_pose)) - what the runners hand on (gp_rank_aggregate_rt)."""
    _lib.check_device()
    B, K, _ = poses.shape
    if poses.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("poses must be float32 or float64")
    poses = poses.contiguous()
    energy = energy.contiguous().float()
    sel = 0
    if selected_num is not None:
        sel = int(selected_num)
    elif ratio is not None:
        sel = max(1, int(K * ratio))
    out = {"sorted_poses": torch.empty_like(poses), "sorted_energy": torch.empty_like(energy),
           "order": torch.empty(B, K, 2, dtype=torch.int32, device=poses.device),
           "avg_pose": torch.empty(B, 7, device=poses.device) if sel > 0 else None}
    if with_rt:
        out["sorted_RTs"] = torch.empty(B, K, 4, 4, dtype=torch.float64, device=poses.device)
        out["avg_RT"] = torch.empty(B, 4, 4, device=poses.device) if sel > 0 else None
        _lib.call("gp_rank_aggregate_rt", B, K, sel, 1 if poses.dtype == torch.float64 else 0, ptr(poses), ptr(energy), ptr(out["sorted_poses"]),
                  ptr(out["sorted_energy"]), ptr(out["order"]), ptr(out["avg_pose"]), ptr(out["sorted_RTs"]), ptr(out["avg_RT"]), stream_ptr())
        return out
    _lib.call("gp_rank_aggregate", B, K, sel, 1 if poses.dtype == torch.float64 else 0, ptr(poses), ptr(energy), ptr(out["sorted_poses"]),
              ptr(out["sorted_energy"]), ptr(out["order"]), ptr(out["avg_pose"]), stream_ptr())
    return out


def sort_poses_by_energy(poses, energy):
    """Same contract as networks/reward.py:131-155 -> (sorted_poses [B,K,9], sorted_energy [B,K,2])."""
    r = rank_aggregate(poses, energy)
    return r["sorted_poses"], r["sorted_energy"]


def cal_average_sRT(poses_sorted_or_unsorted, energy, selected_num):
    """Aggregated 4x4 pose from the top `selected_num` candidates (evaluation_tracking.py:60-77 semantics).
    Takes the UNSORTED 9-D poses + energies (ranking happens in the same launch) -> [B,4,4] f32."""
    r = rank_aggregate(poses_sorted_or_unsorted, energy, selected_num=selected_num)
    return rotation.quat_trans_to_RT(r["avg_pose"])
