"""Phase stamps of the skinny fp32 GEMM at B = 1 shapes (diagnostic build: make EXTRA=-DM3_GEMM_DIAG OBJDIR=build_gdiag
LIB=../tools/_diag_gemm.so; run with M3ASR_LIB=tools/_diag_gemm.so).  Median shader-clock cycles over the work-groups of the LAST
launch: set-up (tile / row descriptors) | issue of the loads | wait for them | MFMAs | LDS reduction + barrier | epilogue | store
drain.  `cold`: rotates through 64 weight matrices (> 256 MB) so that W comes from HBM, as in the 18-layer forward.
usage: diag_gemm_f32.py M N K [ln] [cold]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3m-asr-inference_amd"))
import numpy as np, torch
from m3asr import ops, _lib
M, N, K = (int(v) for v in sys.argv[1:4])
ln, cold = "ln" in sys.argv, "cold" in sys.argv
nw = 96 if cold else 1
a = torch.randn(M, K, device="cuda")
ws = [torch.randn(N, K, device="cuda") * K ** -0.5 for _ in range(nw)]
b = torch.randn(N, device="cuda")
wsum = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda")
for i in range(max(3, nw)):
    ops.linear(a, ws[i % nw], b, out=y, ln_folded=(wsum, None, 1e-12) if ln else None)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(2048 * 8, dtype=np.uint64)
lib.m3_debug_gemm_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.m3_debug_gemm_read(buf.ctypes.data, buf.nbytes) == 0
raw = buf.reshape(2048, 8)
raw = raw[raw[:, 0] > 0]
hw = ((raw[:, 7] >> np.uint64(40)) & np.uint64(0xffff)).astype(np.int64)
xcc = ((raw[:, 7] >> np.uint64(56)) & np.uint64(0xf)).astype(np.int64)
cu = (xcc << 16) | (hw & 0xff00)          # xcc | se_id[15:13] sh_id[12] cu_id[11:8]
per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
d = raw.astype(np.int64)
d[:, 7] = (raw[:, 7] & np.uint64(0xffffffffff)).astype(np.int64)
d[:, :7] = d[:, :7] & 0xffffffffff
names = ["set-up", "load issue", "load wait", "MFMAs", "reduce+barrier", "epilogue", "store drain"]
print("M=%d N=%d K=%d ln=%s cold=%s: %d work-groups" % (M, N, K, ln, cold, len(d)))
for i, n in enumerate(names):
    v = d[:, i + 1] - d[:, i]
    print("  %-15s median %6d  (min %6d, max %6d)" % (n, np.median(v), v.min(), v.max()))
print("  %-15s median %6d; first start -> last end %d cycles" % ("whole block", np.median(d[:, 7] - d[:, 0]), d[:, 7].max() - d[:, 0].min()))
print("  placement: %d work-groups on %d distinct CUs; work-groups per CU: max %d, histogram %s" % (len(d), len(per_cu), per_cu.max(), np.bincount(per_cu)[1:].tolist()))
