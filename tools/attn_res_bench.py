"""ser_attention at the WavLM step's launch shapes, one launch at a time (GPU box): the encoders' form of the call (pre-scaled q, in-kernel
gate, bf16).  Which kernel takes it is decided per process: SER_ATTN_RESIDENT=0 forces the tiled kernel (csrc/attention.hip), the default
lets <= 512-frame launches take the resident-K/V kernel (csrc/attention_res.hip).    bash tools/attn_res_ab.sh"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
H, dh = 16, 64
D = H * dh
shapes = [(8, [499] * 8), (16, [499] * 16), (16, [149] * 16), (16, [300] * 16),
          (16, [499, 160, 333, 250, 480, 64, 401, 499, 200, 450, 310, 97, 499, 380, 275, 499])]
for B, Ts in shapes:
    M, T = sum(Ts), max(Ts)
    for mode in (1, 3):
        dt = torch.bfloat16 if mode == 1 else torch.float16
        qkv = torch.randn(1, M, 3 * D, device=DEV).to(dt)
        xa = torch.randn(1, M, D, device=DEV).to(dt)
        out = torch.empty(1, M, D, dtype=dt, device=DEV)
        offs = torch.tensor(np.concatenate([[0], np.cumsum(Ts)]), dtype=torch.int32, device=DEV)
        table = torch.randn(H, 2 * T - 1, device=DEV)
        cst = torch.randn(H, device=DEV)
        gst = torch.cat([torch.zeros(M, 1), torch.ones(M, 1)], 1).to(DEV).contiguous()
        gw, gcb = (torch.randn(1, 2 * H, dh, device=DEV) * 0.05).to(dt), torch.zeros(H, 4, device=DEV)
        a = L.AttentionArgs()
        a.qkv, a.ld, a.plane_stride, a.q_col, a.k_col, a.v_col, a.B = qkv.data_ptr(), 3 * D, M * 3 * D, 0, D, 2 * D, B
        a.frame_offs, a.table, a.max_frames, a.table_T = offs.data_ptr(), table.data_ptr(), T, T
        a.out, a.ldo, a.out_plane_stride, a.H, a.dh, a.scale, a.mode = out.data_ptr(), D, M * D, H, dh, -1.0, mode
        a.gru_const, a.gate_col = cst.data_ptr(), 3 * D
        a.gate_x, a.gate_x_ld, a.gate_x_plane_stride, a.gate_x_planes = xa.data_ptr(), D, M * D, 1
        a.gate_stat, a.gate_w, a.gate_cb, a.gate_w_plane_stride = gst.data_ptr(), gw.data_ptr(), gcb.data_ptr(), 2 * H * dh
        run = lambda: L.check(L.lib.ser_attention_v(C.byref(a), st))
        for _ in range(5):
            run()
        ts = []
        for r in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        fl = sum(4.0 * H * t * t * dh for t in Ts)
        us = float(np.median(ts))
        print(f"B={B:2d} frames {min(Ts)}..{T} mode {mode}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  ({fl / us / 1e6 / 2500 * 100:4.1f} % of bf16 peak)", flush=True)
