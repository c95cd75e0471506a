"""The clock the chip holds inside ser_gemm's K loop, and where a block spends its life (GPU box, diagnostic build only):

    make -C interspeech_ser_amd/csrc dbg ;  SER_HIP_LIB=$PWD/interspeech_ser_amd/lib/libserhip_dbg.so python tools/gemm_clock.py

MI355X_MICROARCH.md 'DVFS give-back' item 6: in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz, stamped around the
loop after >= 2 s of back-to-back launches on random data, median over workgroups.  Printed with the K-loop rate each tile
reaches at that clock, against the 2.5 PFLOP/s the matrix cores would do at the nominal 2.4 GHz, and with the block anatomy
(entry -> K loop | K loop | epilogue until its last store is acknowledged) of the launches of one WavLM-large layer at the
bench's group shape (8 x 10 s: M = 3992) with their real epilogues."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
dbg = torch.zeros(6 * 65536, dtype=torch.int64, device=DEV)
C.c_void_p.in_dll(L.lib, "ser_gemm_dbg_ptr").value = dbg.data_ptr()
# name, M, N, K, cfg, epilogue flavour, zero operands
CASES = (("fc1 256x256 plain", 3992, 4096, 1024, 3, "plain", 0), ("fc1 256x256 K=4096 plain", 3992, 4096, 4096, 3, "plain", 0),
         ("fc1 256x256 zero operands", 3992, 4096, 4096, 3, "plain", 1), ("whisper fc1 256x256 plain", 12000, 5120, 1280, 3, "plain", 0),
         ("QKV 256x256 (LN, scale)", 3992, 3104, 1024, 3, "qkv", 0), ("FC1 256x256 (LN, GELU)", 3992, 4096, 1024, 3, "fc1", 0),
         ("out-proj 128x128 (residual)", 3992, 1024, 1024, 1, "res", 0), ("FC2 256x128 (residual)", 3992, 1024, 4096, 2, "res", 0))
for name, M, N, K, cfg, flav, zeros in CASES:
    mk = torch.zeros if zeros else torch.randn
    A = mk(1, M, K, device=DEV).to(torch.bfloat16)
    W = (mk(1, N, K, device=DEV) / K ** 0.5).to(torch.bfloat16)
    oa = torch.empty(1, M, N, dtype=torch.bfloat16, device=DEV)
    bias = torch.randn(N, device=DEV)
    keep = [A, W, oa, bias]
    g = L.GemmArgs()
    g.A, g.a_plane_stride, g.lda = A.data_ptr(), M * K, K
    g.W, g.w_plane_stride = W.data_ptr(), N * K
    g.M, g.N, g.K, g.groups, g.mode = M, N, K, 1, 1
    g.out_act, g.ldo_act, g.out_plane_stride = oa.data_ptr(), N, M * N
    g.bias = bias.data_ptr()
    g.tile_cfg = cfg
    if flav in ("qkv", "fc1"):
        groups = K // 64
        stats = torch.rand(M, groups, 2, device=DEV) + 1.0
        stats[:, :, 1] = stats[:, :, 0] ** 2 / 64 + 64.0
        colsum = torch.randn(N, device=DEV)
        g.ln_stats_in, g.ln_groups, g.ln_colsum, g.ln_eps = stats.data_ptr(), groups, colsum.data_ptr(), 1e-5
        keep += [stats, colsum]
        if flav == "fc1":
            g.act = 1
        else:
            g.col_scale, g.col_scale_end = 0.18, 1024
    if flav == "res":
        R = torch.randn(M, N, device=DEV)
        so = torch.empty(M, N // 64, 2, device=DEV)
        sh_in, sh_out = torch.zeros(M, device=DEV), torch.empty(M, device=DEV)
        g.residual, g.ldr = R.data_ptr(), N
        g.out_f32, g.ldo_f32 = R.data_ptr(), N                       # written in place, like the encoder's states
        g.stat_out, g.stat_groups = so.data_ptr(), N // 64
        g.shift_in, g.shift_out, g.shift_const = sh_in.data_ptr(), sh_out.data_ptr(), 0.0
        keep += [R, so, sh_in, sh_out]
    t_end = time.time() + 2.0
    while time.time() < t_end:
        for _ in range(50): L.check(L.lib.ser_gemm(C.byref(g), st))
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): L.check(L.lib.ser_gemm(C.byref(g), st))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    bm, bn = {1: (128, 128), 2: (256, 128), 3: (256, 256)}[cfg]
    nblk = -(-M // bm) * -(-N // bn)
    d = dbg[:2 * nblk].cpu().numpy().reshape(nblk, 2)
    x = dbg[131072:131072 + 4 * nblk].cpu().numpy().reshape(nblk, 4)
    ok = d[:, 1] > 0
    clk = np.median(d[ok, 0] / d[ok, 1]) * 100e6
    loop_s = np.median(d[ok, 1]) * 1e-8
    pro = np.median(x[ok, 0]) / clk * 1e6
    epi = np.median(x[ok, 2] - x[ok, 1]) / clk * 1e6
    per_cu = 2.0 * bm * bn * K / loop_s                                   # FLOP/s of one block's K loop
    peak_cu_at_clk = 2.5e15 / 256 * clk / 2.4e9
    print(f"{name:30s} {us:7.1f} us/launch  clock {clk / 1e9:.2f} GHz  K-loop rate/CU {per_cu / 1e12:.2f} TF/s = "
          f"{100 * per_cu / peak_cu_at_clk:.0f} % of the matrix pipe at that clock ({100 * per_cu * 256 / 2.5e15:.0f} % of nominal) | "
          f"block: entry->loop {pro:5.2f} us, K loop {loop_s * 1e6:6.2f} us, epilogue + store drain {epi:5.2f} us", flush=True)
