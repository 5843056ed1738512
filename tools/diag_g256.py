"""Cycle accounting of expert_gemm_g256_kernel (diagnostic build: make EXTRA=-DM3_G256_DIAG OBJDIR=build_g256diag
LIB=../tools/_diag_g256.so; run with M3ASR_LIB=tools/_diag_g256.so).  The LAST launch of the run is reported (GEMM-2 of the
last operator call; EXP_MODE=1 stops after GEMM-1 is not possible, so GEMM-1 is read by making GEMM-2 a no-op: see -DG256_SKIP2)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3m-asr-inference_amd"))
import numpy as np, torch
from m3asr import ops, _lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D, F, E = 512, 1024, 32
g = torch.Generator().manual_seed(0)
x = torch.randn(S, D, generator=g).cuda()
gate = (torch.randperm(S, generator=g) % E).to(torch.int32).cuda()
w1 = (torch.randn(E, F, D, generator=g) * D ** -0.5).to(torch.bfloat16).cuda()
w2 = (torch.randn(E, D, F, generator=g) * F ** -0.5).to(torch.bfloat16).cuda()
b1, b2 = torch.zeros(E, F).cuda(), torch.zeros(E, D).cuda()
for _ in range(3):
    ops.moe_expert_ffn(x, gate, w1, b1, w2, b2)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.m3_debug_g256_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.m3_debug_g256_read(buf.ctypes.data, buf.nbytes) == 0
d = buf.reshape(4096, 16).astype(np.int64)
d = d[d[:, 0] > 0]
print("records:", len(d), "mode:", np.unique(d[:, 15]))
for mode in np.unique(d[:, 15]):
    m = d[d[:, 15] == mode]
    med = lambda v: int(np.median(v))
    nt = np.maximum(m[:, 3], 1)
    print("MODE %d (%d wave records, %.1f tiles per work-group): cycles PER TILE" % (mode, len(m), float(np.mean(nt))))
    print("  whole kernel / tiles %7d" % med((m[:, 7] - m[:, 0]) / nt))
    print("  tile set-up          %7d" % med(m[:, 4] / nt))
    print("  wait first stages    %7d" % med(m[:, 2] / nt))
    print("  k-loop (whole)       %7d" % med(m[:, 5] / nt))
    print("  fill issue           %7d" % med(m[:, 8] / nt))
    print("  vmcnt wait           %7d" % med(m[:, 9] / nt))
    print("  frag read issue      %7d" % med(m[:, 10] / nt))
    print("  barrier (prep)       %7d" % med(m[:, 11] / nt))
    print("  lgkm wait            %7d" % med(m[:, 12] / nt))
    print("  mfma issue           %7d" % med(m[:, 13] / nt))
    print("  barrier (math)       %7d" % med(m[:, 14] / nt))
    print("  next fills + epilogue%7d" % med(m[:, 6] / nt))
