"""Splits the kernel time of a training run (rocprofv3 --kernel-trace --stats csv) between the hand-written grouping kernels of
libgenpose_hip.so and everything stock torch launches (rocBLAS / MIOpen GEMMs and convolutions, BatchNorm, elementwise, optimiser)."""
import csv
import sys

OURS = ("fps_", "ball_query", "gather_points", "group_points", "pc_step", "rk45", "sa_", "score_", "point_linear", "rank_", "trunk", "cloud_embed", "gp_")
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
groups = {}
for r in rows:
    n = r["Name"]
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
    if any(short.startswith(p) or ("::" + p) in short for p in OURS):
        cls = "genpose_hip: " + short
    elif "Cijk" in n or "gemm" in n.lower() or "rocblas" in n.lower():
        cls = "torch: GEMM (rocBLAS / hipBLASLt)"
    elif "miopen" in n.lower() or "conv" in n.lower() or "batch_norm" in n.lower() or "BatchNorm" in n or "bn_" in n.lower():
        cls = "torch: convolution / BatchNorm (MIOpen, ATen)"
    elif "at::native" in n or "elementwise" in n or "reduce" in n.lower():
        cls = "torch: elementwise / reductions / optimiser (ATen)"
    else:
        cls = "other: " + short[:60]
    g = groups.setdefault(cls, [0, 0.0])
    g[0] += int(r["Calls"])
    g[1] += float(r["TotalDurationNs"])
print(f"kernel time of the profiled run: {tot / 1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
ours = sum(v[1] for k, v in groups.items() if k.startswith("genpose_hip"))
print(f"hand-written grouping / sampling kernels: {ours / 1e6:.1f} ms = {100 * ours / tot:.1f} %   stock torch autograd: {(tot - ours) / 1e6:.1f} ms = {100 * (tot - ours) / tot:.1f} %")
for k, (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    if t / tot >= 0.002:
        print(f"  {k:70s} calls {c:7d}  {t / 1e6:9.2f} ms  {100 * t / tot:5.1f} %")
