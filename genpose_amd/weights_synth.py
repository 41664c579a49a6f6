"""Seeded random weights under the reference's state-dict key names and shapes - stands in for
`ckpt_genpose.pth`, which is not reachable offline.  Output layers (zero-initialised by the reference,
scorenet.py:156-170) are drawn N(0, 0.05) so the score is not identically zero, and BatchNorm statistics are
randomised so BN folding is exercised.  Pure torch-CPU; shared by bench.py, the smoke test, the oracle and the golden
generator so that every party sees identical bytes."""
import math

import torch

# networks/pts_encoder/pointnet2.py:57-66 (ClsMSG_CFG_Light), input_channels = 0
LIGHT_MLPS = [[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]], [[128, 196, 256], [128, 196, 256]],
              [[256, 256, 512], [256, 384, 512]]]


def mlp_specs(mlps=LIGHT_MLPS, input_channels=0):
    """[(level, scale, [Cin+3, c1, c2, c3])] as built by pointnet2.py:173-190 + pointnet2_modules.py:88-90."""
    out = []
    cin = input_channels
    for k, level in enumerate(mlps):
        cout = 0
        for i, m in enumerate(level):
            out.append((k, i, [cin + 3] + list(m)))
            cout += m[-1]
        cin = cout
    return out


def make_state_dict(seed=0, mode="score", params="light"):
    """Random weights with the reference's shapes and key names; output layers (zero-initialised by the
    reference, scorenet.py:156-170) re-drawn N(0,0.05), BN statistics randomised so BN folding is exercised.
    params: the encoder configuration (--pointnet2_params: 'light' | 'dense' | 'lighter', pointnet2.py:47-78)."""
    g = torch.Generator().manual_seed(seed + (0 if mode == "score" else 7919))
    sd = {}

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    if params == "light":
        mlps = LIGHT_MLPS
    else:
        from .weights import ENCODER_CFGS
        mlps = ENCODER_CFGS[params]["mlps"]
    for (k, i, spec) in mlp_specs(mlps):
        for l in range(3):
            cin, cout = spec[l], spec[l + 1]
            p = f"pts_encoder.SA_modules.{k}.mlps.{i}.layer{l}."
            sd[p + "conv.weight"] = rn(cout, cin, 1, 1, std=math.sqrt(2.0 / cin))
            sd[p + "bn.bn.weight"] = 1.0 + 0.1 * rn(cout)
            sd[p + "bn.bn.bias"] = 0.1 * rn(cout)
            sd[p + "bn.bn.running_mean"] = 0.1 * rn(cout)
            sd[p + "bn.bn.running_var"] = (1.0 + 0.1 * rn(cout)).abs() + 0.05
            sd[p + "bn.bn.num_batches_tracked"] = torch.tensor(100)
    q = "pose_score_net."

    def lin(name, cin, cout, std=None):
        std = 1.0 / math.sqrt(cin) if std is None else std
        sd[q + name + ".weight"] = rn(cout, cin, std=std)
        sd[q + name + ".bias"] = rn(cout, std=std)

    lin("pose_encoder.0", 9, 256)
    lin("pose_encoder.2", 256, 256)
    sd[q + "t_encoder.0.W"] = rn(64, std=30.0)
    lin("t_encoder.1", 128, 128)
    for h in ("rot_x", "rot_y", "trans"):
        lin(f"fusion_tail_{h}.0", 1408, 256)
        lin(f"fusion_tail_{h}.2", 256, 3, std=0.05)
    return sd
