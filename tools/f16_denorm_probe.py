"""Does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL operands?  (The f16q mode's lo planes live there: x - fp16(x) of a weight
of size 0.02 is ~1e-5, below fp16's smallest normal 6.1e-5.)  A GEMM of subnormal fp16 A by W = 1 must give the exact sums."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interspeech_ser_amd import _lib as L
M, N, K = 128, 128, 64
A = torch.full((M, K), 2.0 ** -20, dtype=torch.float32)          # fp16 subnormal (smallest normal 2^-14)
A[:, ::2] = 3 * 2.0 ** -24
W = torch.ones((N, K))
W[:, 1::2] = 2.0 ** -17                                           # subnormal on the W side too
a = A.to(torch.float16)[None].contiguous().cuda()
w = W.to(torch.float16)[None].contiguous().cuda()
g = L.GemmArgs()
g.A, g.a_plane_stride, g.lda = a.data_ptr(), M * K, K
g.W, g.w_plane_stride = w.data_ptr(), N * K
g.M, g.N, g.K, g.groups, g.mode = M, N, K, 1, L.MODE_FP16
out = torch.zeros((M, N), device="cuda")
g.out_f32, g.ldo_f32 = out.data_ptr(), N
L.check(L.lib.ser_gemm(C.byref(g), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
ref = a[0].double().cpu() @ w[0].double().cpu().T
print("f16 MFMA subnormal probe: got", float(out[0, 0]), "expected", float(ref[0, 0]),
      "-> subnormals", "KEPT" if torch.allclose(out.cpu().double(), ref, rtol=1e-6, atol=0) else "FLUSHED")
