"""Phase stamps of the FP16M (or any ping-pong) K loop of ser_gemm: one block, units 8 .. 15, every wave's lane 0 (GPU box, diagnostic build):
    make -C interspeech_ser_amd/csrc dbg ; SER_HIP_LIB=$PWD/interspeech_ser_amd/lib/libserhip_dbg.so python tools/gemm_m16_phases.py [cfg]
Columns per unit (shader cycles): DMA issue | fragment reads until lgkmcnt(0) | wait for the next unit's DMAs (late half) | barrier |
MFMA issue | wait for the next unit's DMAs (early half) | barrier."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
dbg = torch.zeros(6 * 65536, dtype=torch.int64, device=DEV)
C.c_void_p.in_dll(L.lib, "ser_gemm_dbg_ptr").value = dbg.data_ptr()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M, N, K = 7984, 1024, 4096
def operand(rows, cols, weight):
    x = torch.randn(rows, cols, device=DEV) * (0.05 if weight else 1.0)
    t = torch.zeros((2, rows, cols), dtype=torch.float16, device=DEV)
    s = torch.zeros((cols // 64, rows), dtype=torch.int32, device=DEV)
    L.check(L.lib.ser_pack_f16m(x.data_ptr(), cols, rows, cols, t.data_ptr(), cols, rows * cols, s.data_ptr(), rows, int(weight), None, st), "pack")
    return t, s
A, As = operand(M, K, False); W, Ws = operand(N, K, True)
oa = torch.zeros((2, M, N), dtype=torch.float16, device=DEV); osc = torch.zeros((N // 64, M), dtype=torch.int32, device=DEV)
g = L.GemmArgs()
g.A, g.a_plane_stride, g.lda, g.W, g.w_plane_stride = A.data_ptr(), M * K, K, W.data_ptr(), N * K
g.M, g.N, g.K, g.groups, g.mode, g.tile_cfg = M, N, K, 1, 6, cfg
g.out_act, g.ldo_act, g.out_plane_stride = oa.data_ptr(), N, M * N
g.a_scale, g.a_scale_ld, g.w_scale, g.w_scale_ld, g.out_scale, g.out_scale_ld = As.data_ptr(), M, Ws.data_ptr(), N, osc.data_ptr(), M
for _ in range(200): L.check(L.lib.ser_gemm(C.byref(g), st))
torch.cuda.synchronize()
d = dbg[262144:262144 + 8 * 8 * 8].cpu().numpy().reshape(8, 8, 8)          # [wave][unit][stamp]
t0 = d[:, 0, 0].min()
print(f"FP16M FC2 shape M={M} N={N} K={K}, tile_cfg {cfg}: block 40, units 8..15; cycles since the first stamp | per-phase deltas")
for w in (0, 4, 1, 5):
    for u in range(8):
        r = d[w, u]
        print(f" wave {w} unit {8 + u} ({'E' if u & 1 else 'H'}): start {r[0] - t0:6d} | issue {r[1] - r[0]:5d} reads {r[2] - r[1]:5d} wait {r[3] - r[2]:5d} barrier {r[4] - r[3]:5d} "
              f"mfma {r[5] - r[4]:5d} wait {r[6] - r[5]:5d} barrier {r[7] - r[6]:5d} | unit {r[7] - r[0]:5d}")
