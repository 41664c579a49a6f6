"""Small rotation helpers on torch tensors (any device / dtype) for the agent's compatibility paths
(`pred_func(return_average_res=True)`, 4x4 RT construction).  The hot path does ranking + aggregation in
csrc/rank.hip; these follow the same published formulas (pytorch3d v0.7.2 semantics, see reference call sites
utils/misc.py:114-148,227-276)."""
import torch
import torch.nn.functional as F


def get_rot_matrix(r6):
    """[N,6] (first two COLUMNS of R) -> [N,3,3]   (utils/misc.py:136)"""
    a1, a2 = r6[..., :3], r6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-1)


def normalize_rotation(r6):
    R = get_rot_matrix(r6)
    return torch.cat([R[..., :, 0], R[..., :, 1]], dim=-1)


def matrix_to_quaternion(m):
    """[N,3,3] -> [N,4] (w,x,y,z): candidate with the largest magnitude component, divisor floored at 0.1."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [m[..., i, j] for i in range(3) for j in range(3)]
    qa = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], dim=-1).clamp_min(0).sqrt()
    cand = torch.stack([
        torch.stack([qa[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, qa[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, qa[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, qa[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * qa[..., None].clamp_min(0.1))
    best = qa.argmax(dim=-1)
    return cand.gather(-2, best[..., None, None].expand(*best.shape, 1, 4)).squeeze(-2)


def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def average_quaternion_batch(Q):
    """[B,n,4] -> [B,4]  (utils/misc.py:227-249, uniform weights)"""
    oq = torch.where(Q[:, :, 0:1] > 0, Q, -Q)
    A = torch.einsum("abi,abk->aik", oq, oq) / Q.shape[1]
    q = torch.linalg.eigh(A)[1][:, :, -1]
    return torch.where(q[:, 0:1] > 0, q, -q)


def pose9_to_RT(pose):
    """[...,9] -> [...,4,4] float64 (evaluation_single.py:325-332).  Device tensors: one launch (gp_pose9_to_rt); host tensors: torch."""
    sh = pose.shape[:-1]
    if pose.is_cuda and pose.dtype in (torch.float32, torch.float64):
        from . import _lib
        p = pose.reshape(-1, 9).contiguous()
        out = torch.empty(p.shape[0], 4, 4, dtype=torch.float64, device=pose.device)
        _lib.call("gp_pose9_to_rt", p.shape[0], 1 if p.dtype == torch.float64 else 0, _lib.ptr(p), _lib.ptr(out), _lib.stream_ptr())
        return out.reshape(sh + (4, 4))
    p = pose.reshape(-1, 9).double()
    RT = torch.eye(4, dtype=torch.float64, device=pose.device).repeat(p.shape[0], 1, 1)
    RT[:, :3, :3] = get_rot_matrix(p[:, :6])
    RT[:, :3, 3] = p[:, 6:]
    return RT.reshape(sh + (4, 4))


def quat_trans_to_RT(avg):
    """[B,7] (w,x,y,z,t) -> [B,4,4].  float32 device tensors: one launch (gp_quat_trans_to_rt); anything else: torch."""
    if avg.is_cuda and avg.dtype == torch.float32:
        from . import _lib
        a = avg.contiguous()
        out = torch.empty(a.shape[0], 4, 4, dtype=torch.float32, device=avg.device)
        _lib.call("gp_quat_trans_to_rt", a.shape[0], _lib.ptr(a), _lib.ptr(out), _lib.stream_ptr())
        return out
    RT = torch.eye(4, dtype=avg.dtype, device=avg.device).repeat(avg.shape[0], 1, 1)
    RT[:, :3, :3] = quaternion_to_matrix(avg[:, :4])
    RT[:, :3, 3] = avg[:, 4:]
    return RT
