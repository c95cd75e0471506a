"""What each stream of the residual GEMMs' epilogue costs (tools/, GPU; round-4 verdict item 3a): the output projection (N = K = 1 024)
and FC2 (N = 1 024, K = 4 096) of WavLM-large at 16 x 10 s (M = 7 984) with the streams the encoder's launches carry --
bias, fp32 residual read (in place), fp32 state store, shifted 16-bit operand copy, row partial sums, shift vector -- switched off one
at a time and all together, one launch at a time (20 back to back between two events, median of 5).
    python tools/gemm_epilogue_whatif.py [M] [mode: bf16 | f16x]
Launch being restated: engine.py `_gemm(pl["ctx"], lay["out"], M, residual=h, out_f32=h, out_act=ha, stat_out=ph, shift=(mx, sh, c))`."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L                     # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 7984
MODE = {"bf16": 1, "f16x": 4}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
PLANES = 2 if MODE == 4 else 1
DT = torch.bfloat16 if MODE == 1 else torch.float16


def operand(rows, cols, weight):
    x = torch.randn(rows, cols, device=DEV) * (0.05 if weight else 1.0)
    t = torch.zeros((PLANES, rows, cols), dtype=DT, device=DEV)
    L.check(L.lib.ser_split_bf16(x.data_ptr(), t.data_ptr(), rows * cols, MODE, rows * cols, torch.cuda.current_stream().cuda_stream), "split")
    return t


def timed(g):
    st = torch.cuda.current_stream().cuda_stream
    ts = []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.check(L.lib.ser_gemm(C.byref(g), st), "ser_gemm")
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / 20)
    return sorted(ts[1:])[2]


def main():
    print(f"M = {M}, mode {MODE}; microseconds per launch (median of 5 x 20 launches), streams: res = fp32 residual read, f32 = fp32 state store, "
          f"act = shifted 16-bit operand copy, stat = row partial sums + shift vector")
    for name, N, K in [("out-proj", 1024, 1024), ("fc2", 1024, 4096)]:
        A, W = operand(M, K, False), operand(N, K, True)
        h = torch.randn(M, N, device=DEV)
        ha = torch.zeros((PLANES, M, N), dtype=DT, device=DEV)
        gD = N // 64
        ph = torch.zeros((M, gD, 2), device=DEV)
        mx, sh = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
        bias = torch.zeros(N, device=DEV)
        rows = []
        for label, res, f32, act, stat in [("full", 1, 1, 1, 1), ("no res", 0, 1, 1, 1), ("no f32", 1, 0, 1, 1), ("no act", 1, 1, 0, 1), ("no stat", 1, 1, 1, 0),
                                           ("res + f32 only", 1, 1, 0, 0), ("act only", 0, 0, 1, 0), ("f32 only", 0, 1, 0, 0), ("act + stat", 0, 0, 1, 1)]:
            g = L.GemmArgs()
            g.A, g.a_plane_stride, g.lda = A.data_ptr(), M * K, K
            g.W, g.w_plane_stride = W.data_ptr(), N * K
            g.M, g.N, g.K, g.groups, g.mode = M, N, K, 1, MODE
            g.bias = bias.data_ptr()
            if res:
                g.residual, g.ldr = h.data_ptr(), N
            if f32:
                g.out_f32, g.ldo_f32 = h.data_ptr(), N
            if act:
                g.out_act, g.ldo_act, g.out_plane_stride = ha.data_ptr(), N, M * N
            if stat:
                g.stat_out, g.stat_groups = ph.data_ptr(), gD
                g.shift_in, g.shift_out, g.shift_const = mx.data_ptr(), sh.data_ptr(), 0.01
            t = timed(g)
            nbytes = 2.0 * PLANES * (M * K + N * K) + 4.0 * M * N * (res + f32) + 2.0 * PLANES * M * N * act + 8.0 * M * gD * stat
            rows.append(f"    {label:16s} {t:7.1f} us   {2.0 * M * N * K / t / 1e6:6.0f} TF/s   {nbytes / 1e6:6.1f} MB algorithmic = {nbytes / t / 1e6:5.2f} TB/s")
            h.normal_()                                   # the in-place residual keeps growing otherwise
        print(f"  {name} N={N} K={K}:")
        print("\n".join(rows))


if __name__ == "__main__":
    main()
