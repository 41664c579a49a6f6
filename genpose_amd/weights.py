"""State-dict -> device weight blocks for the HIP kernels.

Accepts exactly the reference's `model_state_dict` key schema (posenet_agent.py:143-173):
  pts_encoder.SA_modules.{k}.mlps.{i}.layer{l}.conv.weight [Cout,Cin,1,1], ....bn.bn.{weight,bias,running_mean,running_var}
  pose_score_net.{pose_encoder.{0,2}, t_encoder.0.W, t_encoder.1, fusion_tail_{rot_x,rot_y,trans}.{0,2}}.{weight,bias}
so a real `ckpt_genpose.pth` drops in.  Host-side work here is layout only: BatchNorm (eval) folding into the
1x1 conv, input-channel permutation [dx,dy,dz,feat] -> [feat,dx,dy,dz], and the MFMA
fragment packing of gp_pack_weight (include/genpose_hip.h).
"""
import ctypes

import torch

from . import _lib

# networks/pts_encoder/pointnet2.py:24-78 (input_channels = 0, use_xyz=True)
ENCODER_CFGS = {
    "light": dict(npoints=[512, 256, 128, None], radii=[[0.02, 0.04], [0.04, 0.08], [0.08, 0.16], [None, None]],
                  nsamples=[[16, 32], [16, 32], [16, 32], [None, None]],
                  mlps=[[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]], [[128, 196, 256], [128, 196, 256]],
                        [[256, 256, 512], [256, 384, 512]]]),
    "dense": dict(npoints=[512, 256, 128, None], radii=[[0.02, 0.04], [0.04, 0.08], [0.08, 0.16], [None, None]],
                  nsamples=[[32, 64], [16, 32], [8, 16], [None, None]],
                  mlps=[[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]], [[128, 196, 256], [128, 196, 256]],
                        [[256, 256, 512], [256, 384, 512]]]),
    "lighter": dict(npoints=[512, 256, 128, 64, None], radii=[[0.01], [0.02], [0.04], [0.08], [None]],
                    nsamples=[[64], [32], [16], [8], [None]],
                    mlps=[[[32, 32, 64]], [[64, 64, 128]], [[128, 196, 256]], [[256, 256, 512]], [[512, 512, 1024]]]),
}
BN_EPS = 1e-5


def _round16(v):
    return (v + 15) // 16 * 16


def pack_weight(W):
    """W [n_out, k_in] (CPU f32) -> packed MFMA A-operand stream (CPU f32), see gp_common.h."""
    W = W.detach().to(torch.float32).contiguous().cpu()
    n, k = W.shape
    size = _lib.lib().gp_pack_weight_size(n, k)
    out = torch.empty(size, dtype=torch.float32)
    _lib.call("gp_pack_weight", n, k, ctypes.c_void_p(W.data_ptr()), k, ctypes.c_void_p(out.data_ptr()))
    return out


def pad_bias(b):
    b = b.detach().to(torch.float32).cpu()
    out = torch.zeros(_round16(b.numel()), dtype=torch.float32)
    out[: b.numel()] = b
    return out


class SAScale:
    """One (level, scale) of the encoder: three folded layers."""

    def __init__(self, sd, prefix, cin_feat, device):
        self.cin = cin_feat
        self.layers = []
        self.couts = []
        self.hidden_layout = 0  # GP_SA_TAIL_PLAIN
        folded = []
        l = 0
        while f"{prefix}layer{l}.conv.weight" in sd:
            p = f"{prefix}layer{l}."
            W = sd[p + "conv.weight"].detach().double().cpu().reshape(sd[p + "conv.weight"].shape[0], -1)
            gamma, beta = sd[p + "bn.bn.weight"].double().cpu(), sd[p + "bn.bn.bias"].double().cpu()
            mean, var = sd[p + "bn.bn.running_mean"].double().cpu(), sd[p + "bn.bn.running_var"].double().cpu()
            scale = gamma / torch.sqrt(var + BN_EPS)
            Wf = W * scale[:, None]
            bf = beta - mean * scale
            if l == 0:
                if Wf.shape[1] != cin_feat + 3:
                    raise ValueError(f"{p}conv.weight has {Wf.shape[1]} input channels, expected {cin_feat + 3}")
                # hoisted form (csrc/sa_mlp.hip): feature half applied once per source point, xyz half while gathering
                self.w1_feat = Wf[:, 3:].float().contiguous()  # [c1, cin]
                c1p = _round16(Wf.shape[0])
                wxyz = torch.zeros(c1p, 4)
                wxyz[: Wf.shape[0], :3] = Wf[:, :3].float()
                self.wxyz = wxyz.to(device)
                Wf = torch.cat([Wf[:, 3:], Wf[:, :3]], dim=1)  # [dx,dy,dz,feat] -> [feat,dx,dy,dz]
            folded.append((Wf.float(), bf.float()))
            self.couts.append(Wf.shape[0])
            l += 1
        if l != 3:
            raise ValueError(f"{prefix}: the fused SA kernel expects 3-layer shared MLPs, found {l}")
        self._folded_plain = [(W.clone(), b.clone()) for W, b in folded]  # channel order as trained (the fp32 packs below may permute it)
        self._device = device
        self._bf16x3 = None
        c2 = self.couts[1]
        if c2 % 16:
            # hidden width not a multiple of 16 (light level 2: 196): the channels of the last, partly filled 16-channel block go to
            # positions 4 (c % 4) + c / 4 of it (GP_SA_TAIL_SPREAD, genpose_hip.h) - as rows of layer 2 and columns of layer 3 alike, so
            # the network is unchanged - and the chain kernels skip the k-steps of that block that only multiply padding
            self.hidden_layout = 1
            pos = torch.tensor([_lib.lib().gp_sa_tail_position(c2, c) for c in range(c2)], dtype=torch.long)
            (W2, b2), (W3, b3) = folded[1], folded[2]
            W2s, b2s = torch.zeros(_round16(c2), W2.shape[1]), torch.zeros(_round16(c2))
            W2s[pos], b2s[pos] = W2, b2
            W3s = torch.zeros(W3.shape[0], _round16(c2))
            W3s[:, pos] = W3
            folded[1], folded[2] = (W2s, b2s), (W3s, b3)
        self.layers = [(pack_weight(W).to(device), pad_bias(b).to(device)) for W, b in folded]


    def bf16x3_packs(self):
        """Operands of the opt-in split-bf16 kernel (csrc/sa_bf16x3.hip) for layers 2 and 3: every weight as hi = bf16(w) and
        lo = bf16(w - hi), in the fragment order of v_mfma_f32_16x16x32_bf16 with the k order the register chain produces -
        k-block m = channels 32 m .. 32 m + 31, lane group g holds [32m + 4g .. +3] and [32m + 16 + 4g .. +3].
        -> (w2 [KB1][NC2][2][64][8], b2 [32 KB2], w3 [c3/128][KB2][8][2][64][8], b3) as int16 / float32 device tensors."""
        if self._bf16x3 is None:
            (_, _), (W2, b2), (W3, b3) = self._folded_plain
            c1, c2, c3 = self.couts
            nc2 = (c2 + 15) // 16
            kb2 = (nc2 + 1) // 2
            w2 = pack_bf16x3(W2, nc2, c1 // 32)                                   # [KB1][NC2][2][64][8]
            w3 = pack_bf16x3(W3, c3 // 16, kb2)                                   # [KB2][16][2][64][8]
            w3 = torch.stack([w3[:, 8 * h:8 * h + 8] for h in range(c3 // 128)], dim=0).contiguous()  # [half][KB2][8][2][64][8]: slice = (half, k-block)
            b2p = torch.zeros(32 * kb2)
            b2p[:c2] = b2
            dev = self._device
            self._bf16x3 = (w2.to(dev), b2p.to(dev), w3.to(dev), b3.float().contiguous().to(dev))
        return self._bf16x3


def pack_bf16x3(W, n_chunks, k_blocks, chain=True):
    """W [n_out, k_in] f32 -> int16 [k_blocks][n_chunks][2 = hi, lo][64 lanes][8]: the A / B operand fragments of
    v_mfma_f32_16x16x32_bf16 for output chunk nc (16 outputs) and k-block kb (32 inputs), lane l = (n = l % 16, g = l // 16), element e:
    input channel 32 kb + (4 g + e if e < 4 else 16 + 4 g + e - 4) - two consecutive D fragments of the previous layer, as the register
    chain of csrc/sa_bf16x3.hip holds them (chain=False: the natural order 32 kb + 8 g + e, for a layer whose input is not a D fragment).
    Out-of-range outputs / inputs are zero."""
    W = W.detach().float().cpu()
    n_out, k_in = W.shape
    Wp = torch.zeros(16 * n_chunks, 32 * k_blocks)
    Wp[:n_out, :k_in] = W
    hi = Wp.to(torch.bfloat16)
    lo = (Wp - hi.float()).to(torch.bfloat16)
    lanes = torch.arange(64)
    n, g = lanes % 16, lanes // 16
    e = torch.arange(8)
    if chain:
        koff = torch.where(e < 4, 4 * g[:, None] + e[None, :], 16 + 4 * g[:, None] + (e[None, :] - 4))  # [64, 8]
    else:
        koff = 8 * g[:, None] + e[None, :]
    out = torch.empty(k_blocks, n_chunks, 2, 64, 8, dtype=torch.int16)
    for kb in range(k_blocks):
        for nc in range(n_chunks):
            rows = (16 * nc + n)[:, None].expand(64, 8)
            cols = 32 * kb + koff
            out[kb, nc, 0] = hi[rows, cols].view(torch.int16)
            out[kb, nc, 1] = lo[rows, cols].view(torch.int16)
    return out.contiguous()


class EncoderWeights:
    def __init__(self, sd, device, params="light", prefix="pts_encoder."):
        if params not in ENCODER_CFGS:
            raise NotImplementedError(params)
        self.cfg = ENCODER_CFGS[params]
        self.levels = []
        self.z_weights = []
        cin = 0
        for k, level in enumerate(self.cfg["mlps"]):
            scales = [SAScale(sd, f"{prefix}SA_modules.{k}.mlps.{i}.", cin, device) for i in range(len(level))]
            for sc, spec in zip(scales, level):
                if sc.couts != list(spec):
                    raise ValueError(f"SA level {k}: checkpoint layer widths {sc.couts} != config {spec}")
            self.levels.append(scales)
            # combined per-point first-layer weights of the level's scales: z = feat . [W1f_0 ; W1f_1]^T
            self.z_weights.append(pack_weight(torch.cat([sc.w1_feat for sc in scales], dim=0)).to(device) if cin > 0 else None)
            cin = sum(s.couts[-1] for s in scales)
        self.out_dim = cin


class ScoreNetWeights:
    """Device block behind `struct gp_scorenet` (score and energy nets share the layout, posenet.py:58-67)."""

    def __init__(self, sd, device, prefix="pose_score_net."):
        g = lambda k: sd[prefix + k].detach().to(torch.float32).cpu()
        heads = ("rot_x", "rot_y", "trans")
        W1 = torch.cat([g(f"fusion_tail_{h}.0.weight") for h in heads], dim=0)  # [768, 1408] = [pts 1024 | t 128 | pose 256]
        if W1.shape != (768, 1408):
            raise ValueError(f"unexpected fusion_tail first-layer shape {tuple(W1.shape)} (need Rx_Ry_and_T heads, 1024-d pts feature)")
        b1 = torch.cat([g(f"fusion_tail_{h}.0.bias") for h in heads], dim=0)
        W2 = torch.cat([g(f"fusion_tail_{h}.2.weight") for h in heads], dim=0)  # [9, 256]
        b2 = torch.cat([g(f"fusion_tail_{h}.2.bias") for h in heads], dim=0)
        t = {}
        t["w_pose0"] = pack_weight(g("pose_encoder.0.weight"))
        t["b_pose0"] = pad_bias(g("pose_encoder.0.bias"))
        t["w_pose2"] = pack_weight(g("pose_encoder.2.weight"))
        t["b_pose2"] = pad_bias(g("pose_encoder.2.bias"))
        t["w_headx"] = pack_weight(W1[:, 1152:1408].contiguous())
        t["w_out"] = W2.contiguous()
        t["b_out"] = torch.cat([b2, torch.zeros(7)])
        t["fourier_w"] = g("t_encoder.0.W")
        t["w_t1"] = g("t_encoder.1.weight").t().contiguous()  # [in][out]
        t["b_t1"] = g("t_encoder.1.bias")
        t["w_headt"] = W1[:, 1024:1152].t().contiguous()  # [128][768]
        t["w_headp"] = pack_weight(W1[:, :1024].contiguous())
        t["b_head"] = b1.contiguous()
        # transposed packs: backward pass of gp_score_div
        t["w_headx_t"] = pack_weight(W1[:, 1152:1408].t().contiguous())
        t["w_pose2_t"] = pack_weight(g("pose_encoder.2.weight").t().contiguous())
        t["w_pose0_t"] = pack_weight(g("pose_encoder.0.weight").t().contiguous())
        self.tensors = {k: v.to(device) for k, v in t.items()}
        self.struct = _lib.GpScoreNet(**{k: v.data_ptr() for k, v in self.tensors.items()})
        self._raw = {"pose0": g("pose_encoder.0.weight"), "pose2": g("pose_encoder.2.weight"), "headx": W1[:, 1152:1408].contiguous()}
        self._device = device
        self._bf16x3 = None

    def bf16x3_packs(self):
        """Operands of the opt-in split-bf16 PC step (csrc/trunk_bf16x3.hip): the three dense layers of the trunk as hi / lo bf16 pairs in
        the fragment order of v_mfma_f32_16x16x32_bf16 (pack_bf16x3) -> (w_pose0 [1][16][2][64][8], w_pose2 [8][16]..., w_headx [8][48]...)."""
        if self._bf16x3 is None:
            r = self._raw
            self._bf16x3 = (pack_bf16x3(r["pose0"], 16, 1, chain=False).to(self._device), pack_bf16x3(r["pose2"], 16, 8).to(self._device),
                            pack_bf16x3(r["headx"], 48, 8).to(self._device))
        return self._bf16x3

    def ref(self):
        return ctypes.byref(self.struct)
