"""Calibration (GPU box): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the shapes ser_gemm runs, plain
bf16 x bf16 -> bf16 with no epilogue, one launch at a time -- the same measurement as bench.py's roofline leg.  Not used by the
product; it tells how much of the distance to the 2.5 PFLOP/s peak is the shape and how much is ser_gemm."""
import numpy as np, torch
DEV = "cuda:0"
SHAPES = [("qkv  8utt", 3992, 3104, 1024), ("out  8utt", 3992, 1024, 1024), ("fc1  8utt", 3992, 4096, 1024), ("fc2  8utt", 3992, 1024, 4096),
          ("qkv 16utt", 7984, 3104, 1024), ("out 16utt", 7984, 1024, 1024), ("fc1 16utt", 7984, 4096, 1024), ("fc2 16utt", 7984, 1024, 4096),
          ("conv1 8utt", 127992, 512, 1536), ("big square", 8192, 8192, 8192)]
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=DEV, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, w.t())
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            torch.matmul(a, w.t())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ms = float(np.median(ts))
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
