"""Tuning build only: which pc_step workgroups share a CU and how their lifetimes overlap.  python scratch/occupancy.py B K"""
import sys, ctypes; sys.path.insert(0, '.')
import torch, numpy as np
from collections import defaultdict
from genpose_amd import _lib
from genpose_amd.scorenet import ScoreNetHIP
from genpose_amd.samplers import PCSampler
from genpose_amd.weights_synth import make_state_dict
B, K = int(sys.argv[1]), int(sys.argv[2])
net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
cvec = torch.randn(B, 768, device="cuda"); cen = torch.zeros(B, 3, device="cuda")
smp = PCSampler(net, B, K, 20, "cuda", use_graph=False)
x0 = torch.randn(B * K, 9, device="cuda") * 50
for _ in range(2): smp.run(cvec, cen, x0)
torch.cuda.synchronize()
l = ctypes.CDLL(_lib.SO_PATH)
buf = (ctypes.c_ulonglong * 4096)()
assert l.gp_debug_wg_stamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(1024, 4).astype(np.int64)[: smp.nblocks]
hw, xcc = a[:, 0], a[:, 1] & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7  # gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
groups = defaultdict(list)
for i in range(len(a)): groups[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append(i)
print("workgroups", len(a), "distinct CUs", len(groups), "per-CU counts", np.bincount([len(v) for v in groups.values()]))
dur = a[:, 3] - a[:, 2]
print("duration cycles: mean %.0f min %d max %d" % (dur.mean(), dur.min(), dur.max()))
n = 0
for k, v in sorted(groups.items()):
    if len(v) > 1 and n < 12:
        t0 = min(a[i, 2] for i in v)
        print(k, [(i, int(a[i, 2] - t0), int(a[i, 3] - t0)) for i in v]); n += 1
