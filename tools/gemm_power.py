"""Board power beside each encoder-layer GEMM shape of WavLM-large (M = 7 984, bf16 unless a mode is given), launched back to back for a few
seconds each (tools/, GPU; run under tools/power_beside.sh, which samples rocm-smi and matches the wall-clock stamps printed here):
    bash tools/power_beside.sh python tools/gemm_power.py [seconds] [mode: bf16|f16x|f16m]
The step runs at the board's cap; this shows what each kernel draws at the rate it sustains (compare tools/energy_probe.hip: a loop of
the same shape with ideal overlap sustains ~1 600 TF/s at ~1 200 W)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L                     # noqa: E402
_argv, sys.argv = sys.argv, sys.argv[:1]                      # (the module reads its own M from argv at import)
import gemm_f16m_bench as GB                                  # noqa: E402  (operand packing)
sys.argv = _argv

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
MODE = {"bf16": 1, "f16": 3, "f16x": 4, "f16m": 6}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
M = 7984


def main():
    st = torch.cuda.current_stream().cuda_stream
    for name, N, K, act in GB.SHAPES:
        A, As = GB.operand(M, K, MODE, False)
        W, Ws = GB.operand(N, K, MODE, True)
        planes = 2 if MODE in (2, 4, 6) else 1
        oa = torch.zeros((planes, M, N), dtype=torch.float16, device=GB.DEV)
        osc = torch.zeros((N // 64, M), dtype=torch.int32, device=GB.DEV)
        g = L.GemmArgs()
        g.A, g.a_plane_stride, g.lda = A.data_ptr(), M * K, K
        g.W, g.w_plane_stride = W.data_ptr(), N * K
        g.M, g.N, g.K, g.groups, g.mode, g.act = M, N, K, 1, MODE, act
        g.out_act, g.ldo_act, g.out_plane_stride = oa.data_ptr(), N, M * N
        if MODE == 6:
            g.a_scale, g.a_scale_ld, g.w_scale, g.w_scale_ld = As.data_ptr(), M, Ws.data_ptr(), N
            g.out_scale, g.out_scale_ld = osc.data_ptr(), M
        for _ in range(50):
            L.check(L.lib.ser_gemm(C.byref(g), st), "ser_gemm")
        torch.cuda.synchronize()
        t0 = time.time()
        n, us = 0, 0.0
        while time.time() - t0 < SECONDS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                L.check(L.lib.ser_gemm(C.byref(g), st), "ser_gemm")
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 200
            n += 200
        t2 = time.time()
        print(f"{name}_N{N}_K{K} first launch 0 | {n} launches, last 200: {us:7.1f} us = {2.0 * M * N * K / us / 1e6:7.1f} TF/s   wall {t0:.3f} .. {t0:.3f} .. {t2:.3f}", flush=True)


if __name__ == "__main__":
    main()
