"""Four ser_attention launches (bf16, dh=64, T=499): plain, bias, plain-prescaled, bias-prescaled -- for rocprofv3 --pmc."""
import ctypes as C, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
B, H, dh, T = 16, 16, 64, 499
D = H * dh; M = B * T
qkv = torch.randn(1, M, 3 * D, device=DEV).to(torch.bfloat16)
out = torch.empty(1, M, D, dtype=torch.bfloat16, device=DEV)
offs = torch.arange(0, M + 1, T, dtype=torch.int32, device=DEV)
table = torch.randn(H, 2 * T - 1, device=DEV); gate = torch.rand(M, H, device=DEV)
for bias, scale in ((0, dh ** -0.5), (1, dh ** -0.5), (0, -1.0), (1, -1.0)):
    L.check(L.lib.ser_attention(qkv.data_ptr(), 3 * D, M * 3 * D, 0, D, 2 * D, offs.data_ptr(), B, T,
                                table.data_ptr() if bias else None, T if bias else 0, gate.data_ptr() if bias else None,
                                out.data_ptr(), D, M * D, H, dh, scale, 1, 0, None, None, None, 0, st))
    torch.cuda.synchronize()
