"""Where a wave of ser_attention spends a key tile (GPU box, debug build only):

    make -C interspeech_ser_amd/csrc dbg ;  SER_HIP_LIB=$PWD/interspeech_ser_amd/lib/libserhip_dbg.so python tools/attn_phases.py [T] [bias]

The debug build stamps s_memtime at six points of every tile for the four waves of block 100 and writes them to the
buffer whose address this tool plants in the library's `ser_attn_dbg_ptr`.  Phases: issue of the next tile's global loads +
LDS reads + S MFMAs | V reads + max | exp / PV | wait for the staged loads + LDS writes | barrier."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
T = int(sys.argv[1]) if len(sys.argv) > 1 else 499
bias = int(sys.argv[2]) if len(sys.argv) > 2 else 1
MODE = int(sys.argv[3]) if len(sys.argv) > 3 else 1        # 1 bf16, 3 fp16, 5 fp16q, 4 fp16x, 2 fp32x (two-plane modes need scale <= 0)
PL = 2 if MODE in (2, 4, 5) else 1
DT = torch.bfloat16 if MODE in (1, 2) else torch.float16
B, H, dh = 8, 16, 64
D = H * dh
M = B * T
qkv = torch.randn(PL, M, 3 * D, device=DEV).to(DT)
out = torch.empty(PL, M, D, dtype=DT, device=DEV)
offs = torch.arange(0, M + 1, T, dtype=torch.int32, device=DEV)
table = torch.randn(H, 2 * T - 1, device=DEV); gate = torch.rand(M, H, device=DEV)
dbg = torch.zeros(4 * 64 * 6, dtype=torch.int64, device=DEV)
C.c_void_p.in_dll(L.lib, "ser_attn_dbg_ptr").value = dbg.data_ptr()
def run():
    L.check(L.lib.ser_attention(qkv.data_ptr(), 3 * D, M * 3 * D, 0, D, 2 * D, offs.data_ptr(), B, T,
                                table.data_ptr() if bias else None, T if bias else 0, gate.data_ptr() if bias else None,
                                out.data_ptr(), D, M * D, H, dh, -1.0 if PL == 2 else dh ** -0.5, MODE, 0, None, None, None, 0, st))
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"kernel: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (debug build)")
d = dbg.cpu().numpy().reshape(4, 64, 6)
nkt = (T + 63) // 64
names = ["loads+reads+S", "V reads+max", "exp/PV", "vmcnt+ds_write", "barrier"]
for w in range(4):
    print(f"wave {w}: prologue-issue -> tile 0 start {d[w, 0, 0] - d[w, 63, 0]} ticks")
    for kt in range(nkt):
        t = d[w, kt]
        ph = [int(t[i + 1] - t[i]) for i in range(5)]
        gap = int(d[w, kt + 1, 0] - t[5]) if kt + 1 < nkt else 0
        print(f"  tile {kt:2d}: " + "  ".join(f"{n} {v:5d}" for n, v in zip(names, ph)) + f"   total {int(t[5] - t[0]):5d}")
print('prologue stamps (ticks from entry): T known | tile-0 loads issued | Q+gate issued | bias copied | K/V written | barrier passed')
for w in range(4): print('   wave', w, [int(x) for x in d[w, 61]])
print('entry -> end of loop in 10 ns units (s_memrealtime):', [int(d[w, 62, 2]) for w in range(4)])
print('entry -> end of loop (ticks):', [int(d[w, 62, 1] - d[w, 62, 0]) for w in range(4)], ' entry -> prologue issued:', [int(d[w, 63, 0] - d[w, 62, 0]) for w in range(4)])
tot = d[0, nkt - 1, 5] - d[0, 0, 0]
print(f"wave 0, {nkt} tiles: {tot} ticks")
