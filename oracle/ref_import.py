"""ORACLE — TEST INFRASTRUCTURE ONLY; runs ONLY in the build container (needs /root/reference).

Imports the reference's own Python (networks/posenet_agent.py etc.) on CPU so that golden vectors
can be generated from it (oracle/gen_golden.py).  Recipe = SURVEY.md App. B:

  * stub modules for packages the image lacks (oracle/shims/: ipdb, tensorboardX, cv2, torchvision,
    pytorch3d -> oracle/rot.py restatement),
  * `pointnet2_cuda` replaced by the CPU C restatement (oracle/pn2_ops.c) because the reference's CUDA
    extension cannot be built here (no nvcc / NVIDIA device),
  * torch.cuda.{Int,Float}Tensor aliased to the CPU constructors (pointnet2_utils.py:26-27 etc.),
  * sys.argv set before the first reference import (pointnet2.py:13 parses argv at import).

Nothing here is used on the GPU box; /root/reference does not exist there.
"""
import os
import sys

REF = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_loaded = {}


def available():
    return os.path.isdir(os.path.join(REF, "networks"))


def load(sampler_mode="ode", extra_argv=()):
    """Returns a namespace with the reference modules; idempotent."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError("reference tree not present (this only runs in the build container)")
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    repo = os.path.dirname(_HERE)
    for p in (repo, os.path.join(_HERE, "shims"), REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    # order matters: shims first, then reference
    sys.path.remove(os.path.join(_HERE, "shims"))
    sys.path.insert(0, os.path.join(_HERE, "shims"))
    import torch

    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    from oracle import pn2_oracle

    sys.modules["pointnet2_cuda"] = pn2_oracle.as_pointnet2_cuda_module()
    saved_argv = sys.argv
    sys.argv = ["ref", "--device", "cpu", "--sampler_mode", sampler_mode, "--pose_mode", "rot_matrix"] + list(extra_argv)
    try:
        import types

        ns = types.SimpleNamespace()
        from configs.config import get_config
        from networks.posenet_agent import PoseNet
        from networks.posenet import GFObjectPose
        from networks.reward import sort_poses_by_energy
        from networks.gf_algorithms import sde as ref_sde
        from networks.gf_algorithms import samplers as ref_samplers
        from networks.pts_encoder.pointnet2_utils.pointnet2 import pointnet2_utils as ref_pn2_utils
        from utils import misc as ref_misc
        from utils import sgpa_utils as ref_sgpa
        from utils import tracking_utils as ref_tracking

        ns.get_config = get_config
        ns.PoseNet = PoseNet
        ns.GFObjectPose = GFObjectPose
        ns.sort_poses_by_energy = sort_poses_by_energy
        ns.sde = ref_sde
        ns.samplers = ref_samplers
        ns.pn2_utils = ref_pn2_utils
        ns.misc = ref_misc
        ns.sgpa = ref_sgpa
        ns.tracking = ref_tracking
        ns.cfg = get_config()
    finally:
        sys.argv = saved_argv
    _loaded["ns"] = ns
    return ns
