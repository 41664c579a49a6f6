"""Gate of the split-bf16 experiment: the 128-196-256 grouping level, fp32 MFMA kernels against the split-bf16 kernel -
time per launch at 320 / 640 clouds and the error of each against fp64.  python scratch/bf16x3_gate.py > profiles/r5_bf16x3_gate.txt"""
import sys; sys.path.insert(0, ".")
import torch
from genpose_amd import _lib, synth
from genpose_amd._lib import ptr, stream_ptr
from genpose_amd.encoder import Pointnet2EncoderHIP
from genpose_amd.weights_synth import make_state_dict
sys.path.insert(0, "tests")

sd = make_state_dict(0, "score")
enc = Pointnet2EncoderHIP(sd, "cuda")
for B in (320, 640):
    pts = torch.from_numpy(synth.make_batch(B)).cuda()
    _, ws = enc.forward(pts, return_intermediates=True)
    torch.cuda.synchronize()
    k, n, npnt = 2, 256, 128
    scales = enc.w.levels[k]
    z = ws["z"][k]
    zstride = sum(sc.couts[0] for sc in scales)
    out = torch.empty_like(ws["feat"][k])
    xyz, new_xyz = ws["new_xyz"][1], ws["new_xyz"][2]
    st = stream_ptr()
    def run(which):
        off = zoff = 0
        for i, sc in enumerate(scales):
            ns = enc.cfg["nsamples"][k][i]
            (w1, b1), (w2, b2), (w3, b3) = sc.layers
            if which == "bf16x3":
                w2s, b2s, w3s, b3s = sc.bf16x3_packs()
                _lib.call("gp_sa_pre_mlp_max_bf16x3", B, n, npnt, ns, 128, 196, 256, ptr(xyz), ptr(new_xyz), ptr(ws["bq"][k][i]), ptr(z), zstride, zoff,
                          ptr(sc.wxyz), ptr(b1), ptr(w2s), ptr(b2s), ptr(w3s), ptr(b3s), ptr(out), out.shape[-1], off, st)
            else:
                _lib.call("gp_sa_pre_mlp_max_layout", sc.hidden_layout, B, n, npnt, ns, 128, 196, 256, ptr(xyz), ptr(new_xyz), ptr(ws["bq"][k][i]), ptr(z),
                          zstride, zoff, ptr(sc.wxyz), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(out), out.shape[-1], off, st)
            off += 256; zoff += 128
    res = {}
    for which in ("f32", "bf16x3"):
        for _ in range(3): run(which)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(which)
        e1.record(); torch.cuda.synchronize()
        res[which] = (e0.elapsed_time(e1) / 20 * 1e3, out.clone())
    flop = B * 128 * (16 + 32) * 2 * (128 * 196 + 196 * 256)  # executed by the two scales (layers 2 and 3)
    t32, tbf = res["f32"][0], res["bf16x3"][0]
    print(f"{B} clouds, level 2 (both scales, {flop / 1e9:.1f} GFLOP in layers 2-3): fp32 MFMA {t32:.0f} us ({flop / t32 / 1e6:.1f} TFLOP/s)   "
          f"split bf16 {tbf:.0f} us ({flop / tbf / 1e6:.1f} TFLOP/s of fp32-equivalent work)   speed-up {t32 / tbf:.2f}x")
    if B == 320:
        from test_gpu_bf16x3 import _level2_fp64
        ref = _level2_fp64(enc, ws, 4)
        sc_ = float(ref.abs().max())
        for which in ("f32", "bf16x3"):
            d = (res[which][1][:4].double().cpu() - ref).abs()
            print(f"   {which:7s} vs fp64 (4 clouds): max |err| / feature scale {float(d.max()) / sc_:.2e}, rms {float(d.pow(2).mean().sqrt()) / sc_:.2e}")
