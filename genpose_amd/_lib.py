"""ctypes binding of libgenpose_hip.so (the C ABI of include/genpose_hip.h).

There is NO CPU fallback: if the library is missing or a call fails, the caller gets an exception.
`lib()` only dlopens (safe without a GPU); `check_device()` verifies the device is gfx950.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GENPOSE_HIP_LIB: tuning only (a variant built with GP_BUILD_TAG=<tag> beside the shipped library, genpose_amd/build.py)
SO_PATH = os.environ.get("GENPOSE_HIP_LIB") or os.path.join(_HERE, "lib", "libgenpose_hip.so")

c_int, c_float, c_void_p, c_int64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64
P = c_void_p
PLAN_SHARED = 0x200     # GP_PLAN_SHARED: the RK45 driver's shared-chunk plan (16 | or 48 |), see include/genpose_hip.h
PLAN_FLAGS = 0x300
PLAN_HEADSPLIT = 0x100  # GP_PLAN_HEADSPLIT (include/genpose_hip.h): OR-ed onto a 16-row tile plan = three workgroups per tile, one head each


class GpScoreNet(ctypes.Structure):
    """struct gp_scorenet (include/genpose_hip.h)."""
    _fields_ = [(n, c_void_p) for n in ("w_pose0", "b_pose0", "w_pose2", "b_pose2", "w_headx", "w_out", "b_out", "fourier_w",
                                        "w_t1", "b_t1", "w_headt", "w_headp", "b_head", "w_headx_t", "w_pose2_t", "w_pose0_t")]


NETP = ctypes.POINTER(GpScoreNet)

# name -> argtypes ; every function returns int (0 ok / negative GP_E*), except gp_pack_weight_size (int64)
SIGNATURES = {
    "gp_version": [],
    "gp_device_arch": [ctypes.c_char_p, c_int],
    "gp_arith_default": [],
    "gp_furthest_point_sampling": [c_int, c_int, c_int, P, P, P, P],
    "gp_furthest_point_sampling_arith": [c_int, c_int, c_int, c_int, P, P, P, P],
    "gp_ball_query_arith": [c_int, c_int, c_int, c_int, c_float, c_int, P, P, P, P],
    "gp_three_nn_arith": [c_int, c_int, c_int, c_int, P, P, P, P, P],
    "gp_three_interpolate_arith": [c_int, c_int, c_int, c_int, c_int, P, P, P, P, P],
    "gp_fps_chain_arith": [c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), P, P, P, P, P, P, P, P],
    "gp_ball_query_msg_arith": [c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_int, P, P, P, P, P],
    "gp_gather_points": [c_int, c_int, c_int, c_int, P, P, P, P],
    "gp_gather_points_grad": [c_int, c_int, c_int, c_int, P, P, P, P],
    "gp_ball_query": [c_int, c_int, c_int, c_float, c_int, P, P, P, P],
    "gp_group_points": [c_int, c_int, c_int, c_int, c_int, P, P, P, P],
    "gp_group_points_grad": [c_int, c_int, c_int, c_int, c_int, P, P, P, P],
    "gp_three_nn": [c_int, c_int, c_int, P, P, P, P, P],
    "gp_three_interpolate": [c_int, c_int, c_int, c_int, P, P, P, P, P],
    "gp_three_interpolate_grad": [c_int, c_int, c_int, c_int, P, P, P, P, P],
    "gp_fps_chain": [c_int, c_int, c_int, ctypes.POINTER(c_int), P, P, P, P, P, P, P, P],
    "gp_ball_query_msg": [c_int, c_int, c_int, c_float, c_int, c_float, c_int, P, P, P, P, P],
    "gp_sa_mlp_max": [c_int] * 8 + [P] * 10 + [P, c_int, c_int, P],
    "gp_point_linear": [c_int, c_int, c_int, P, P, P, P],
    "gp_sa_pre_mlp_max": [c_int] * 7 + [P, P, P, P, c_int, c_int] + [P] * 7 + [c_int, c_int, P],
    "gp_sa_pre_mlp_max_layout": [c_int] * 8 + [P, P, P, P, c_int, c_int] + [P] * 7 + [c_int, c_int, P],
    "gp_sa_tail_position": [c_int, c_int],
    "gp_sa_pre_mlp_max_bf16x3": [c_int] * 7 + [P, P, P, P, c_int, c_int] + [P] * 7 + [c_int, c_int, P],
    "gp_pack_weight_size": [c_int, c_int],
    "gp_pack_weight": [c_int, c_int, P, c_int, P],
    "gp_score_tile_rows": [c_int],
    "gp_cloud_embed": [c_int, NETP, P, P, P],
    "gp_time_embed": [c_int, NETP, P, P, P],
    "gp_score_eval": [c_int, c_int, NETP, P, P, P, P, c_int, P, P],
    "gp_pc_step": [c_int, c_int, c_int, c_int, NETP] + [P] * 11 + [P],
    "gp_roi_to_cloud": [c_int, c_int, c_int, c_int, P, P, P, c_float, c_float, c_float, c_float, P, P, P, P],
    "gp_cloud_sample": [c_int, c_int, c_int, P, P, P, P, P],
    "gp_score_div": [c_int, c_int, NETP, P, P, P, P, P, P, P, P],
    "gp_energy_score": [c_int, c_int, NETP, P, P, P, P, P, P, P],
    "gp_pc_tile_rows": [c_int, c_int, c_int],
    "gp_pc_layout": [c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)],
    "gp_pc_step_plan": [c_int, c_int, c_int, c_int, c_int, c_int, c_int, NETP] + [P] * 12 + [c_int, P],
    "gp_score_eval_plan": [c_int, c_int, c_int, NETP, P, P, P, P, c_int, P, P],
    "gp_pc_layout_bf16x3": [c_int, c_int, c_int, ctypes.POINTER(c_int)],
    "gp_pc_step_bf16x3": [c_int] * 5 + [P] * 18 + [P],
    "gp_pc_step_grouped": [c_int, c_int, c_int, c_int, c_int, NETP] + [P] * 11 + [P],
    "gp_pc_step_coupled": [c_int, c_int, c_int, c_int, c_int, NETP] + [P] * 12 + [P],
    "gp_rk45_state_bytes": [],
    "gp_rk45_state_layout": [ctypes.POINTER(c_int64), c_int],
    "gp_rk45_set_dense": [P, P, c_int, P, P],
    "gp_rk45_phase": [c_int, c_int, c_int, NETP, P, P, P, P, P, P, P, P, P, c_int] + [ctypes.c_double] * 5 + [c_int, c_int, P, P],
    "gp_rk45_phase_grouped": [c_int, c_int, c_int, c_int, NETP, P, P, P, P, P, P, P, P, P, c_int] + [ctypes.c_double] * 5 + [c_int, c_int, P, P],
    "gp_rk45_plan_rows": [c_int, c_int, c_int, c_int],
    "gp_plan_headsplit_pays": [c_int],
    "gp_rk45_plan_rows_unshared": [c_int, c_int, c_int, c_int],
    "gp_rk45_partials_count": [c_int, c_int, c_int, c_int, c_int],
    "gp_rk45_phase_model": [c_int, c_int, P, c_int, c_int, c_int, c_int, NETP, P, P, P, P, P, P, P, P, P, c_int] + [ctypes.c_double] * 5
                           + [c_int, c_int, P, P, c_int, P],
    "gp_rk45_set_dense_grouped": [c_int, P, P, c_int, P, P],
    "gp_rk45_phase_ragged": [c_int, c_int, P, c_int, P, c_int, c_int, c_int, NETP, P, P, P, P, P, P, P, P, P, c_int] + [ctypes.c_double] * 5
                            + [c_int, c_int, P, P],
    "gp_time_embed_strided": [c_int, c_int, c_int64, NETP, P, P, P],
    "gp_rank_aggregate": [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P],
    "gp_rank_aggregate_rt": [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P],
    "gp_pose9_to_rt": [c_int, c_int, P, P, P],
    "gp_quat_trans_to_rt": [c_int, P, P, P],
}
OPTIONAL = set()

_lib = None


class GenposeHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  torch's bundled libamdhip64.so.7 must be the process's HIP runtime (shared streams/pointers)
        if not os.path.exists(SO_PATH):
            raise GenposeHipError(
                f"{SO_PATH} is missing - build it with `python -m genpose_amd.build` (hipcc, gfx950). "
                "genpose_amd has no CPU fallback.")
        l = ctypes.CDLL(SO_PATH)
        for name, args in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                if name in OPTIONAL:
                    continue
                raise GenposeHipError(f"{SO_PATH} does not export {name} (stale build?)")
            fn.argtypes = args
            fn.restype = c_int64 if name in ("gp_pack_weight_size", "gp_rk45_state_bytes") else c_int
        _lib = l
    return _lib


_ERR = {-1: "GP_EINVAL (bad size / null pointer / unsupported shape)", -2: "GP_ELAUNCH (HIP launch failed)",
        -3: "GP_EARCH (device is not gfx950)"}


def check(rc, what):
    if rc != 0:
        raise GenposeHipError(f"{what} failed: {_ERR.get(rc, rc)}")


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


_device_checked = False


def check_device():
    """Raise unless the current HIP device is gfx950 (MI355X)."""
    global _device_checked
    if _device_checked:
        return
    import torch
    if not torch.cuda.is_available():
        raise GenposeHipError("genpose_amd needs an MI355X (gfx950) device; torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.init()
    buf = ctypes.create_string_buffer(64)
    rc = lib().gp_device_arch(buf, 64)
    if rc != 0:
        raise GenposeHipError(f"genpose_amd needs an MI355X (gfx950) device; found '{buf.value.decode()}' ({_ERR.get(rc, rc)})")
    _device_checked = True


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return c_void_p(torch.cuda.current_stream().cuda_stream)
