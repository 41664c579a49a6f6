"""python scratch/ubench/run.py - cycles per MFMA beside NV VALU instructions (see valu_mfma.hip); the library is built by scratch/ubench/build.sh."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libub.so"))
lib.ub_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(256 * 4 * 256, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
iters = 4000
names = {0: "v_fma_f32", 1: "v_pk_fma_f32", 2: "v_max_f32 v,v", 3: "bf16 MFMA + v_fma_f32", 4: "v_mul_f32", 5: "v_add_f32", 6: "v_pk_mul_f32", 7: "v_pk_add_f32", 8: "v_max_i32", 9: "v_med3_f32", 10: "v_mov_b32", 11: "v_max3_f32", 12: "v_cndmask_b32", 13: "v_max_f32 0,v"}
def run(nv, kind, w):
    lds = {1: 100, 2: 70, 3: 50, 4: 36}[w] * 1024
    blocks = 256 * w
    def go():
        rc = lib.ub_launch(nv, kind, out.data_ptr(), iters, blocks, lds, st)
        assert rc == 0, rc
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); go(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    return ms * 1e6 / (iters * 16 * w)   # ns per MFMA per SIMD
base = run(0, 0, 2)
print(f"fp32 MFMA alone: {base:.2f} ns per MFMA per SIMD (two waves per SIMD)")
for kind in sorted(names):
    r4, r8 = run(4, kind, 2), run(8, kind, 2)
    b = run(0, kind, 2)
    print(f"   {names[kind]:24s} NV=4: {r4:6.2f}  NV=8: {r8:6.2f}   per instruction: {(r8 - b) / 8:5.2f} ns = {(r8 - b) / 8 * 2.4:4.1f} cycles at 2.4 GHz")
