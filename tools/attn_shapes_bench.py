"""ser_attention on the four encoders' shapes in the production call form (q pre-scaled, WavLM gate as fused columns),
bf16 and f16, one utterance group per launch.  SER_HIP_LIB selects an alternative build for same-device A/B."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
SHAPES = [("wavlm-L 8x499", 8, 16, 64, 499, True), ("wavlm-L 16x499", 16, 16, 64, 499, True), ("hubert-XL 8x499", 8, 16, 80, 499, False),
          ("xlsr-2b 4x499", 4, 16, 120, 499, False), ("whisper 8x1500", 8, 20, 64, 1500, False)]
modes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,3").split(",")]
for mode in modes:
    dt = torch.float16 if mode == 3 else torch.bfloat16
    planes = 2 if mode == 2 else 1
    for name, B, H, dh, T, bias in SHAPES:
        D, M = H * dh, B * T
        ld = 3 * D + (32 if bias else 0)
        qkv = (torch.randn(planes, M, ld, device=DEV) * 0.5).to(dt)
        out = torch.empty(planes, M, D, dtype=dt, device=DEV)
        offs = torch.arange(0, M + 1, T, dtype=torch.int32, device=DEV)
        table = torch.randn(H, 2 * T - 1, device=DEV) if bias else None
        cst = torch.rand(H, device=DEV) + 0.5 if bias else None

        def run():
            L.check(L.lib.ser_attention(qkv.data_ptr(), ld, M * ld, 0, D, 2 * D, offs.data_ptr(), B, T,
                                        table.data_ptr() if bias else None, T if bias else 0, None, out.data_ptr(), D, M * D, H, dh,
                                        -1.0, mode, 3 * D, cst.data_ptr() if bias else None, None, None, 0, st))
        for _ in range(3):
            run()
        ts = []
        for r in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        us = float(np.median(ts))
        fl = 4.0 * B * H * T * T * dh
        print(f"mode={mode} {name:16s}: {us:7.1f} us  {fl/us/1e6:6.1f} TF/s   lib={os.path.basename(L.LIB_PATH)}", flush=True)
