"""The clock the chip holds inside ser_gemm's K loop (GPU box, diagnostic build only):

    hipcc ... -DSER_GEMM_DBG -c gemm.hip -> lib/libserhip_dbg.so ;  SER_HIP_LIB=<that> python tools/gemm_clock.py

MI355X_MICROARCH.md 'DVFS give-back' item 6: in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz, stamped around the
loop after >= 2 s of back-to-back launches on random data, median over workgroups.  Printed with the K-loop rate each tile
reaches at that clock, against the 2.5 PFLOP/s the matrix cores would do at the nominal 2.4 GHz."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
dbg = torch.zeros(2 * 65536, dtype=torch.int64, device=DEV)
C.c_void_p.in_dll(L.lib, "ser_gemm_dbg_ptr").value = dbg.data_ptr()
for name, M, N, K, cfg, zeros in (("fc1 256x256", 3992, 4096, 1024, 3, 0), ("fc1 256x256 K=4096", 3992, 4096, 4096, 3, 0),
                                  ("whisper fc1 256x256", 12000, 5120, 1280, 3, 0), ("fc2 256x128", 3992, 1024, 4096, 2, 0),
                                  ("out-proj 128x128", 3992, 1024, 1024, 1, 0), ("fc1 256x256, zero operands", 3992, 4096, 4096, 3, 1)):
    A = (torch.zeros if zeros else torch.randn)(1, M, K, device=DEV).to(torch.bfloat16)
    W = ((torch.zeros if zeros else torch.randn)(1, N, K, device=DEV) / K ** 0.5).to(torch.bfloat16)
    oa = torch.empty(1, M, N, dtype=torch.bfloat16, device=DEV)
    g = L.GemmArgs()
    g.A, g.a_plane_stride, g.lda = A.data_ptr(), M * K, K
    g.W, g.w_plane_stride = W.data_ptr(), N * K
    g.M, g.N, g.K, g.groups, g.mode = M, N, K, 1, 1
    g.out_act, g.ldo_act, g.out_plane_stride = oa.data_ptr(), N, M * N
    g.tile_cfg = cfg
    t_end = time.time() + 2.0
    n = 0
    while time.time() < t_end:
        for _ in range(50): L.check(L.lib.ser_gemm(C.byref(g), st))
        torch.cuda.synchronize(); n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): L.check(L.lib.ser_gemm(C.byref(g), st))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    bm, bn = {1: (128, 128), 2: (256, 128), 3: (256, 256)}[cfg]
    nblk = -(-M // bm) * -(-N // bn)
    d = dbg[:2 * nblk].cpu().numpy().reshape(nblk, 2)
    ok = d[:, 1] > 0
    clk = np.median(d[ok, 0] / d[ok, 1]) * 100e6
    loop_s = np.median(d[ok, 1]) * 1e-8
    cus = min(nblk, 256)
    per_cu = 2.0 * bm * bn * K / loop_s                                   # FLOP/s of one block's K loop
    peak_cu_at_clk = 2.5e15 / 256 * clk / 2.4e9
    print(f"{name:28s} {us:7.1f} us/launch  K loop {loop_s * 1e6:6.2f} us  clock {clk / 1e9:.2f} GHz  "
          f"K-loop rate/CU {per_cu / 1e12:.2f} TF/s = {100 * per_cu / peak_cu_at_clk:.0f} % of the matrix pipe at that clock "
          f"({100 * per_cu * 256 / 2.5e15:.0f} % of nominal)", flush=True)
