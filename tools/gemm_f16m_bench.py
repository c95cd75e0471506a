"""ser_gemm per numerics mode on the encoder-layer shapes of WavLM-large at 16 x 10 s (M = 7 984), one launch at a time (tools/, GPU):
    python tools/gemm_f16m_bench.py [M]
bf16 / f16 (1 product), f16x (3 products on fp16 hi + lo), f16m (fp16 + scaled e4m3 cross terms).  Each launch writes the operand copy of
its own mode (+ GELU for FC1) like the encoder's launches do; the deferred LayerNorm, residual and statistics are left out."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from interspeech_ser_amd import _lib as L                     # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 7984
SHAPES = [("qkv", 3072, 1024, 0), ("out", 1024, 1024, 0), ("fc1", 4096, 1024, 1), ("fc2", 1024, 4096, 0)]
MODES = [("bf16", 1), ("f16", 3), ("f16x", 4), ("f16m", 6)]


def operand(rows, cols, mode, weight):
    x = torch.randn(rows, cols, device=DEV) * (0.05 if weight else 1.0)
    if mode == 6:
        t = torch.zeros((2, rows, cols), dtype=torch.float16, device=DEV)
        s = torch.zeros((cols // 64, rows), dtype=torch.int32, device=DEV)
        L.check(L.lib.ser_pack_f16m(x.data_ptr(), cols, rows, cols, t.data_ptr(), cols, rows * cols, s.data_ptr(), rows, int(weight), None,
                                    torch.cuda.current_stream().cuda_stream), "pack")
        return t, s
    planes = 2 if mode in (2, 4) else 1
    dt = torch.bfloat16 if mode in (1, 2) else torch.float16
    t = torch.zeros((planes, rows, cols), dtype=dt, device=DEV)
    L.check(L.lib.ser_split_bf16(x.data_ptr(), t.data_ptr(), rows * cols, mode, rows * cols, torch.cuda.current_stream().cuda_stream), "split")
    return t, None


def main():
    st = torch.cuda.current_stream().cuda_stream
    print(f"M = {M}; microseconds per launch (median of 5 x 20 launches)")
    for name, N, K, act in SHAPES:
        row = []
        for mname, mode in MODES:
            A, As = operand(M, K, mode, False)
            W, Ws = operand(N, K, mode, True)
            planes = 2 if mode in (2, 4, 6) else 1
            oa = torch.zeros((planes, M, N), dtype=torch.float16, device=DEV)
            osc = torch.zeros((N // 64, M), dtype=torch.int32, device=DEV)
            g = L.GemmArgs()
            g.A, g.a_plane_stride, g.lda = A.data_ptr(), M * K, K
            g.W, g.w_plane_stride = W.data_ptr(), N * K
            g.M, g.N, g.K, g.groups, g.mode, g.act = M, N, K, 1, mode, act
            if mode == 6:
                g.tile_cfg = int(os.environ.get("SER_BENCH_CFG", "0"))        # 0 auto, 2 = 256x128, 3 = 256x256
            if mode in (1, 3):
                g.tile_cfg = int(os.environ.get("SER_BENCH_CFG1", "0"))       # single-plane modes: 4 = 256x256 on four waves
            g.out_act, g.ldo_act, g.out_plane_stride = oa.data_ptr(), N, M * N
            if mode == 6:
                g.a_scale, g.a_scale_ld, g.w_scale, g.w_scale_ld = As.data_ptr(), M, Ws.data_ptr(), N
                g.out_scale, g.out_scale_ld = osc.data_ptr(), M
            ts = []
            for rep in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    L.check(L.lib.ser_gemm(C.byref(g), st), "ser_gemm")
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1000 / 20)
            t = sorted(ts[1:])[2]
            row.append(f"{mname} {t:7.1f} us ({2.0 * M * N * K / t / 1e6:6.0f} TF/s)")
        print(f"  {name:4s} N={N:5d} K={K:5d}: " + "   ".join(row))


if __name__ == "__main__":
    main()
