"""Phase stamps of the LDS-DMA GEMM (diagnostic build: make EXTRA=-DM3_DMA_DIAG OBJDIR=build_diag LIB=../tools/_diag_dma.so, run with
M3ASR_LIB=tools/_diag_dma.so M3_DMA_MIN_ROWS=512).  Prints, over the work-groups of one launch, the median shader-clock cycles
of: set-up, first fill (prologue), k-loop, accumulators -> LDS image, epilogue sweep, store drain; and the spread of start times.
usage: diag_gemm_dma.py M N K [resid]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3m-asr-inference_amd"))
import numpy as np, torch
from m3asr import ops, _lib
M, N, K = (int(v) for v in sys.argv[1:4])
resid = len(sys.argv) > 4
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda")
r = torch.randn(M, N, device="cuda") if resid else None
y = torch.empty(M, N, device="cuda")
yb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if resid else None
for _ in range(5):
    ops.linear(a, w, b, out=y, resid=r, copy_bf16=yb)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.m3_debug_dma_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.m3_debug_dma_read(buf.ctypes.data, buf.nbytes) == 0
nwg = ((M + 127) // 128 + 7) // 8 * 8 * ((N + 127) // 128)
d = buf.reshape(4096, 8)[:min(nwg, 4096)].astype(np.int64)
d = d[d[:, 0] > 0]
names = ["set-up", "first fill", "k-loop", "acc -> image", "epilogue sweep", "store drain"]
print("M=%d N=%d K=%d resid=%s: %d work-groups with stamps" % (M, N, K, resid, len(d)))
for i, n in enumerate(names):
    v = d[:, i + 1] - d[:, i]
    print("  %-16s median %7d cycles  (min %7d, max %7d)" % (n, np.median(v), v.min(), v.max()))
tot = d[:, 6] - d[:, 0]
print("  %-16s median %7d cycles; start spread %d cycles, end spread %d cycles (shader clock; 100 MHz realtime span %d ticks)" % (
    "whole block", np.median(tot), d[:, 0].max() - d[:, 0].min(), d[:, 6].max() - d[:, 6].min(), d[:, 7].max() - d[:, 7].min()))
