"""Gate of the split-bf16 experiment: grouping levels 1 (64-64/96-128) and 2 (128-196-256), fp32 MFMA kernels against the
split-bf16 kernel - time per launch pair at 320 / 640 clouds and the error of each against fp64; then the whole encoder pass.
    python scratch/bf16x3_gate.py > profiles/r5_bf16x3_gate.txt"""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from genpose_amd import _lib, synth
from genpose_amd._lib import ptr, stream_ptr
from genpose_amd.encoder import Pointnet2EncoderHIP
from genpose_amd.weights_synth import make_state_dict

sd = make_state_dict(0, "score")
enc = Pointnet2EncoderHIP(sd, "cuda")
encb = Pointnet2EncoderHIP(sd, "cuda", precision="bf16x3")


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B in (320, 640):
    pts = torch.from_numpy(synth.make_batch(B)).cuda()
    _, ws = enc.forward(pts, return_intermediates=True)
    torch.cuda.synchronize()
    for k, n, npnt in ((1, 512, 256), (2, 256, 128)):
        scales = enc.w.levels[k]
        z = ws["z"][k]
        zstride = sum(sc.couts[0] for sc in scales)
        out = torch.empty_like(ws["feat"][k])
        xyz, new_xyz = ws["new_xyz"][k - 1], ws["new_xyz"][k]
        st = stream_ptr()

        def run(which):
            off = zoff = 0
            for i, sc in enumerate(scales):
                ns = enc.cfg["nsamples"][k][i]
                (w1, b1), (w2, b2), (w3, b3) = sc.layers
                c1, c2, c3 = sc.couts
                if which == "bf16x3":
                    w2s, b2s, w3s, b3s = sc.bf16x3_packs()
                    _lib.call("gp_sa_pre_mlp_max_bf16x3", B, n, npnt, ns, c1, c2, c3, ptr(xyz), ptr(new_xyz), ptr(ws["bq"][k][i]), ptr(z), zstride, zoff,
                              ptr(sc.wxyz), ptr(b1), ptr(w2s), ptr(b2s), ptr(w3s), ptr(b3s), ptr(out), out.shape[-1], off, st)
                else:
                    _lib.call("gp_sa_pre_mlp_max_layout", sc.hidden_layout, B, n, npnt, ns, c1, c2, c3, ptr(xyz), ptr(new_xyz), ptr(ws["bq"][k][i]), ptr(z),
                              zstride, zoff, ptr(sc.wxyz), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(out), out.shape[-1], off, st)
                off += c3; zoff += c1
        res = {}
        for which in ("f32", "bf16x3"):
            t = timeit(lambda: run(which))
            res[which] = (t, out.clone())
        flop = sum(B * npnt * ns * 2 * (sc.couts[0] * sc.couts[1] + sc.couts[1] * sc.couts[2]) for sc, ns in zip(scales, enc.cfg["nsamples"][k]))
        t32, tbf = res["f32"][0], res["bf16x3"][0]
        print(f"{B} clouds, level {k} (both scales, {flop / 1e9:.1f} GFLOP in layers 2-3): fp32 MFMA {t32:.0f} us ({flop / t32 / 1e6:.1f} TFLOP/s)   "
              f"split bf16 {tbf:.0f} us ({flop / tbf / 1e6:.1f} TFLOP/s of fp32-equivalent work)   speed-up {t32 / tbf:.2f}x")
        if B == 320:
            from test_gpu_bf16x3 import _level_fp64
            ref = _level_fp64(enc, ws, 4, k)
            sc_ = float(ref.abs().max())
            for which in ("f32", "bf16x3"):
                d = (res[which][1][:4].double().cpu() - ref).abs()
                print(f"   {which:7s} vs fp64 (4 clouds): max |err| / feature scale {float(d.max()) / sc_:.2e}, rms {float(d.pow(2).mean().sqrt()) / sc_:.2e}")
    t32 = timeit(lambda: enc.encode(pts)[0], 10)
    tbf = timeit(lambda: encb.encode(pts)[0], 10)
    f32, fbf = enc.encode(pts)[0], encb.encode(pts)[0]
    dev = float((fbf - f32).abs().max()) / float(f32.abs().max())
    print(f"{B} clouds, WHOLE encoder pass (one graph replay): fp32 {t32 / 1e3:.3f} ms   levels 1-2 on split bf16 {tbf / 1e3:.3f} ms   ({t32 / tbf:.2f}x); "
          f"final features differ by {dev:.2e} of their scale")
