"""Depth -> cloud pre-processing: GPU time per frame vs the CPU oracle.  python scratch/bench_pre.py [n_inst]"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from genpose_amd import preprocess as pp, synth
from oracle import preprocess_oracle as po
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
depth, masks, rois, cls = synth.golden_depth_frame()
rep = (n + 3) // 4
keep = [0, 1, 2, 3]
masks = np.concatenate([masks[:, :, keep]] * rep, axis=2)[:, :, :n]; rois = np.concatenate([rois[keep]] * rep)[:n]; cls = np.concatenate([cls[keep]] * rep)[:n]
d2c = pp.DepthToClouds()
dd, mm = torch.as_tensor(depth).cuda(), torch.as_tensor(masks).cuda().to(torch.uint8)
d2c.full_clouds(dd, mm, rois); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): pcl, cnt, dc = d2c.full_clouds(dd, mm, rois)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
valid = int(cnt.sum())
byts = n * 65536 * 3 + valid * 12
t = time.time(); out = d2c(dd, mm, rois, cls); torch.cuda.synchronize(); t_full = time.time() - t
t = time.time(); po.frame_clouds(depth, masks, rois, cls, po.REAL_INTRINSICS); t_cpu = time.time() - t
print(f"n_inst={n}: roi_to_cloud {ms*1e3:.1f} us/frame (incl. host map + launch), {byts/ms/1e6:.1f} GB/s algorithmic ({byts/1e6:.2f} MB), "
      f"end-to-end call {t_full*1e3:.2f} ms, CPU oracle {t_cpu*1e3:.1f} ms")
