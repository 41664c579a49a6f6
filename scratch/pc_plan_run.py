"""One PC sampler chain on a given launch plan (for rocprofv3 runs): python scratch/pc_plan_run.py <tile> [G=10] [steps=30]"""
import sys

import torch

sys.path.insert(0, ".")
from genpose_amd.samplers import PCSampler  # noqa: E402
from genpose_amd.scorenet import ScoreNetHIP  # noqa: E402
from genpose_amd.weights_synth import make_state_dict  # noqa: E402

tile = int(sys.argv[1])
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
B1, K = 64, 50
net = ScoreNetHIP(make_state_dict(0, "score"), "cuda")
smp = PCSampler(net, G * B1, K, n, "cuda", use_graph=False, groups=G, tile=tile)
cvec = torch.randn(G * B1, 768, device="cuda")
centre = torch.randn(G * B1, 3, device="cuda")
x0 = torch.randn(G * B1 * K, 9, device="cuda") * 50
for _ in range(2):
    smp.run(cvec, centre, x0)
torch.cuda.synchronize()
print("ran", smp.kernel_name, G * B1 * K, "rows")
