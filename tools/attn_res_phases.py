"""Where the four waves of one block of the resident-K/V attention kernel spend their time (GPU box, debug build only):

    make -C interspeech_ser_amd/csrc dbg ; SER_HIP_LIB=$PWD/interspeech_ser_amd/lib/libserhip_dbg.so python tools/attn_res_phases.py [T] [B]

The debug build stamps s_memtime for every wave of block 100 at: entry | frame offsets known | register loads requested | registers +
tile 0 landed | all DMA pieces issued | gates done (loop start) | the start of every (query block, key tile) | the end of each query block."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
T = int(sys.argv[1]) if len(sys.argv) > 1 else 499
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H, dh, mode = 16, 64, 1
D, M = H * dh, B * T
qkv = torch.randn(1, M, 3 * D, device=DEV).to(torch.bfloat16)
xa = torch.randn(1, M, D, device=DEV).to(torch.bfloat16)
out = torch.empty(1, M, D, dtype=torch.bfloat16, device=DEV)
offs = torch.arange(0, M + 1, T, dtype=torch.int32, device=DEV)
table = torch.randn(H, 2 * T - 1, device=DEV)
cst = torch.randn(H, device=DEV)
gst = torch.cat([torch.zeros(M, 1), torch.ones(M, 1)], 1).to(DEV).contiguous()
gw, gcb = (torch.randn(1, 2 * H, dh, device=DEV) * 0.05).to(torch.bfloat16), torch.zeros(H, 4, device=DEV)
dbg = torch.zeros(8 * 64 * 6, dtype=torch.int64, device=DEV)
C.c_void_p.in_dll(L.lib, "ser_attn_dbg_ptr").value = dbg.data_ptr()
a = L.AttentionArgs()
a.qkv, a.ld, a.plane_stride, a.q_col, a.k_col, a.v_col, a.B = qkv.data_ptr(), 3 * D, M * 3 * D, 0, D, 2 * D, B
a.frame_offs, a.table, a.max_frames, a.table_T = offs.data_ptr(), table.data_ptr(), T, T
a.out, a.ldo, a.out_plane_stride, a.H, a.dh, a.scale, a.mode = out.data_ptr(), D, M * D, H, dh, -1.0, mode
a.gru_const, a.gate_col = cst.data_ptr(), 3 * D
a.gate_x, a.gate_x_ld, a.gate_x_plane_stride, a.gate_x_planes = xa.data_ptr(), D, M * D, 1
a.gate_stat, a.gate_w, a.gate_cb, a.gate_w_plane_stride = gst.data_ptr(), gw.data_ptr(), gcb.data_ptr(), 2 * H * dh
run = lambda: L.check(L.lib.ser_attention_v(C.byref(a), st))
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"kernel: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (debug build), B={B} T={T}")
d = dbg.cpu().numpy()[: 8 * 64].reshape(8, 64)
nkt = (T + 63) // 64
t0 = d[:, 0].min()
print("ticks from the block's first entry:  entry | T known | loads requested | regs + tile 0 landed | DMA issued | loop start | block 0 done | block 1 done")
NW = 4
t0 = d[:NW, 0].min()
for w in range(NW):
    print(f"  wave {w}: " + " ".join(f"{int(d[w, i] - t0):6d}" for i in (0, 1, 2, 3, 4, 5, 33, 35)))
for j in range(2):
    print(f"round {j} (64 queries per wave): ticks per 64-key tile (start to next start)")
    for w in range(NW):
        s = [int(d[w, 8 + j * 8 + kt]) for kt in range(nkt)] + [int(d[w, 32 + 2 * j])]
        if s[0] == 0:
            continue
        print(f"  wave {w}: " + " ".join(f"{s[i + 1] - s[i]:5d}" for i in range(nkt)) + f"   total {s[-1] - s[0]:6d}   first tile starts at {s[0] - t0:6d}")

print("inside half-tile 4 of round 1 (32 keys x 64 queries), ticks: K+bias reads, init FMAs | 8 S MFMAs | V reads requested | 2 row maxima | branch | exp + PV MFMAs | row sums")
for w in range(NW):
    v = [int(d[w, i]) for i in range(40, 48)]
    if v[0]:
        print(f"  wave {w}: " + " ".join(f"{v[i + 1] - v[i]:5d}" for i in range(7)) + f"   total {v[7] - v[0]:5d}")
