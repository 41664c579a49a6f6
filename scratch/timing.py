"""Tuning build only (GP_TIMING=1 python -m genpose_amd.build --force): phase timestamps of pc_step block 0."""
import sys, ctypes; sys.path.insert(0, '.')
import torch, numpy as np
from genpose_amd import _lib
from genpose_amd.scorenet import ScoreNetHIP
from genpose_amd.samplers import PCSampler
from genpose_amd.weights_synth import make_state_dict
B, K = int(sys.argv[1]), int(sys.argv[2])
net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
cvec = torch.randn(B, 768, device="cuda"); cen = torch.zeros(B, 3, device="cuda")
smp = PCSampler(net, B, K, 100, "cuda", use_graph=False)
x0 = torch.randn(B * K, 9, device="cuda") * 50
for _ in range(3): smp.run(cvec, cen, x0)
torch.cuda.synchronize()
l = _lib.lib(); l.gp_debug_timestamps.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 256)()
assert l.gp_debug_timestamps(buf) == 0
ts = np.array(buf, dtype=np.uint64).reshape(8, 32).astype(np.int64)
names = {0: "start", 19: "loads+gn", 1: "prologue done", 2: "trunk in", 3: "L1 done(+bar)", 4: "L2 mfma+epi", 5: "L2 barrier", 6: "h0 start", 7: "h0 mfma", 8: "h0 epi",
         9: "h1 start", 10: "h1 mfma", 11: "h1 epi", 12: "h2 start", 13: "h2 mfma", 14: "h2 epi", 16: "trunk out", 17: "end"}
for w in range(8):
    if ts[w, 0] == 0: continue
    t0 = ts[w, 0]
    print(f"wave {w}: " + "  ".join(f"{names[i]}={ts[w, i] - t0}" for i in sorted(names) if ts[w, i] > 0))
