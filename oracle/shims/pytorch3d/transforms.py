"""Import shim (build container only) -> oracle/rot.py."""
from oracle.rot import rotation_6d_to_matrix, matrix_to_rotation_6d, quaternion_to_matrix, matrix_to_quaternion  # noqa: F401
def euler_angles_to_matrix(*a, **k):
    raise NotImplementedError("pytorch3d shim: euler modes are outside the hot path")
