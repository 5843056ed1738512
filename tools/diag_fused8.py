"""Phase stamps of the fused fp8 expert kernel (diagnostic build: make EXTRA=-DM3_FUSED_DIAG OBJDIR=build_f8diag LIB=../tools/_diag_f8.so,
run with M3ASR_LIB=tools/_diag_f8.so).  Prints, over the waves of one launch, median shader-clock cycles per phase.
usage: diag_fused8.py S E [F]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3m-asr-inference_amd"))
import numpy as np, torch
from m3asr import ops, _lib
from m3asr.plan import quantize_fp8_rows
S, E = int(sys.argv[1]), int(sys.argv[2])
F = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
D = 512
g = torch.Generator().manual_seed(0)
x = torch.randn(S, D, generator=g).cuda()
gate = (torch.randperm(S, generator=g) % E).to(torch.int32).cuda()
w1 = torch.randn(E, F, D, generator=g) * D ** -0.5
w2 = torch.randn(E, D, F, generator=g) * F ** -0.5
b1, b2 = (torch.randn(E, F, generator=g) * 0.1).cuda(), (torch.randn(E, D, generator=g) * 0.1).cuda()
q1, s1 = quantize_fp8_rows(w1, dims=(2,))
q2, s2 = quantize_fp8_rows(w2, dims=(2,))
q1, s1, q2, s2 = q1.cuda(), s1.cuda(), q2.cuda(), s2.cuda()
for _ in range(4):
    ops.moe_expert_ffn(x, gate, q1, b1, q2, b2, w1_scale=s1, w2_scale=s2, h_scale=0.05)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.m3_debug_fused8_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.m3_debug_fused8_read(buf.ctypes.data, buf.nbytes) == 0
d = buf.reshape(4096, 8).astype(np.int64)
d = d[d[:, 7] > 0]
names = ["barrier wait", "Y stores", "tile end (next X, small operands)", "GEMM-1 steps", "GEMM-2 steps", "step loop total", "prologue", "whole kernel"]
print("S=%d E=%d F=%d: %d waves with stamps (%d work-groups)" % (S, E, F, len(d), len(d) // 4))
for i, n in enumerate(names):
    v = d[:, i]
    print("  %-36s median %7d  min %7d  max %7d" % (n, np.median(v), v.min(), v.max()))
