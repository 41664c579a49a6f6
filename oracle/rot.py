"""ORACLE — TEST INFRASTRUCTURE ONLY.

Restatement of the four pytorch3d.transforms functions GenPose calls.  pytorch3d (pinned v0.7.2 by
the reference's README.md:56,68-74) is a third-party dependency that is NOT vendored under
/root/reference and is not installed in this image, so the functions are restated here from
their published algorithm (Zhou et al. 6-D rotation / standard quaternion algebra):

  rotation_6d_to_matrix : Gram-Schmidt, b1=normalize(a1), b2=normalize(a2-(b1.a2)b1), b3=b1xb2, rows=(b1,b2,b3)
  matrix_to_rotation_6d : first two rows, flattened
  quaternion_to_matrix  : real-first (w,x,y,z), two_s = 2/|q|^2
  matrix_to_quaternion  : four candidates, pick argmax |q_i|, divisor floored at 0.1 (0.7.2 does not
                          standardise the sign; every GenPose consumer re-orients to w>0, misc.py:242,248)

Call sites in the reference: utils/misc.py:133,136 (get_rot_matrix), networks/posenet_agent.py:453,
runners/evaluation_tracking.py:67,75, utils/sgpa_utils.py:940,948.  No reference test pins these
-> PARITY UNPINNED at the pytorch3d level; pinned only by our own fixtures.
"""
import torch
import torch.nn.functional as F


def rotation_6d_to_matrix(d6):
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)  # eps 1e-12
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def matrix_to_rotation_6d(matrix):
    batch_dim = matrix.size()[:-2]
    return matrix[..., :2, :].clone().reshape(batch_dim + (6,))


def quaternion_to_matrix(quaternions):
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    positive_mask = x > 0
    ret[positive_mask] = torch.sqrt(x[positive_mask])
    return ret


def matrix_to_quaternion(matrix):
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(
        torch.stack(
            [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1
        )
    )
    quat_by_rijk = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    return quat_candidates[F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(batch_dim + (4,))
