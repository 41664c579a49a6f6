"""CPU (no GPU): the committed trained checkpoints (tests/golden/trained/, made by scratch/train_synth.py with the reference-pinned training
step on the device) are reference-layout files that the ORACLE can run - and they are trained: on held-out synthetic instances the oracle's ODE
sampler lands near the ground-truth pose far more often than chance.  (The GPU side of the same files: tests/test_gpu_trained_regime.py.)"""
import os

import numpy as np
import torch

from oracle import genpose_oracle as go

HERE = os.path.dirname(os.path.abspath(__file__))
CKPT = {m: os.path.join(HERE, "golden", "trained", f"ckpt_{m}.pth") for m in ("score", "energy")}


def _sd(mode):
    ck = torch.load(CKPT[mode], map_location="cpu")
    assert set(ck) >= {"model_state_dict", "clock"} and ck["clock"]["step"] > 1000
    return {k: v.float() for k, v in ck["model_state_dict"].items()}


def test_schema_and_inference_agent_accepts_them():
    from genpose_amd.config import get_config
    from genpose_amd.posenet_agent import PoseNet
    ref = go.make_state_dict(0, "score")
    for mode in ("score", "energy"):
        sd = _sd(mode)
        assert list(sd) == list(ref) or set(sd) == set(ref)
        assert all(sd[k].shape == ref[k].shape for k in ref)
        a = PoseNet(get_config(device="cpu", posenet_mode=mode))
        a.load_ckpt(model_dir=CKPT[mode], model_path=True, load_model_only=True)  # weight packing / BN folding on the host: no device needed


def test_oracle_on_trained_weights_finds_the_pose():
    from genpose_amd import synth
    torch.set_num_threads(4)
    sd = _sd("score")
    B, K = 6, 8
    d = synth.posed_batch(range(2_000_000, 2_000_000 + B))
    pts = torch.from_numpy(d["pts"])
    prior = torch.randn(B * K, 9, generator=torch.Generator().manual_seed(3))
    pred, _, nfev = go.pred_func(sd, pts, pts.mean(dim=1), K, "ode", prior, T0=0.55)
    assert 60 < nfev < 600 and torch.isfinite(pred).all()
    # translation: every candidate within 5 cm of the truth (chance: the prior at T0 = 0.55 has sigma 1.08 m); rotation of the asymmetric
    # categories' best candidate within 20 degrees for most instances (chance: 3 % per candidate)
    t_err = np.linalg.norm(pred[:, :, 6:].numpy() - d["t"][:, None, :], axis=-1)
    assert np.median(t_err) < 0.02 and t_err.max() < 0.10, (np.median(t_err), t_err.max())
    R = go.get_rot_matrix(pred.reshape(B * K, 9)[:, :6].float()).numpy().reshape(B, K, 3, 3)
    y_err = np.degrees(np.arccos(np.clip(np.einsum("bki,bi->bk", R[:, :, :, 1], d["R"][:, :, 1]), -1, 1)))  # the object's y axis: defined for every category
    assert np.mean(y_err.min(axis=1) < 20.0) >= 0.5, y_err.min(axis=1)
