"""What a store-only / copy kernel of the GEMM epilogues' size costs on this box (GPU): bounds the fixed cost of a
launch whose tile results all leave the chip in one burst."""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for mb in (1, 8, 16, 32, 64, 256, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    tf = t(lambda: x.zero_()); tc = t(lambda: y.copy_(x))
    print(f"{mb:5d} MB  fill {tf:7.1f} us = {mb/1024/1024*1e6*1.048576/tf:6.2f} TB/s   copy {tc:7.1f} us = {2*mb*1.048576e6/tc/1e12*1e0:6.2f} TB/s (r+w)", flush=True)
