"""ser_attention timing anatomy (GPU box): T sweep, with/without the WavLM bias."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
B, H, dh = 16, 16, 64
D = H * dh
for mode in (1, 2):
    planes = 2 if mode == 2 else 1
    for T in (128, 256, 499, 998, 1500):
        M = B * T
        qkv = torch.randn(planes, M, 3 * D, device=DEV).to(torch.bfloat16)
        out = torch.empty(planes, M, D, dtype=torch.bfloat16, device=DEV)
        offs = torch.arange(0, M + 1, T, dtype=torch.int32, device=DEV)
        table = torch.randn(H, 2 * T - 1, device=DEV); gate = torch.rand(M, H, device=DEV)
        res = []
        for bias in (0, 1):
            def run():
                L.check(L.lib.ser_attention(qkv.data_ptr(), 3 * D, M * 3 * D, 0, D, 2 * D, offs.data_ptr(), B, T,
                                            table.data_ptr() if bias else None, T if bias else 0,
                                            gate.data_ptr() if bias else None, out.data_ptr(), D, M * D, H, dh,
                                            dh ** -0.5, mode, 0, None, None, None, 0, st))
            for _ in range(3): run()
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            res.append(np.median(ts))
        fl = 4.0 * B * H * T * T * dh
        print(f"mode={mode} T={T:5d}: plain {res[0]:7.1f}us ({fl/res[0]/1e6:6.1f} TF)  bias {res[1]:7.1f}us ({fl/res[1]/1e6:6.1f} TF)", flush=True)
