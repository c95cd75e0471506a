"""ser_attention at the WavLM step's shapes (GPU box), gate pre-activations read from two extra columns of the packed projection against
computed in the kernel from the layer input's operand copy (ser_attention_args.gate_x): the kernel alone, one launch at a time.
    python tools/attn_gate_bench.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
DEV = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
NAMES = {1: "BF16", 3: "FP16", 4: "FP16X", 2: "FP32X"}
H, dh = 16, 64
D = H * dh
for B, T in ((8, 499), (16, 499)):
    M = B * T
    for mode in (1, 4):
        dt = torch.bfloat16 if mode in (1, 2) else torch.float16
        planes = 1 if mode in (1, 3) else 2
        for form in ("columns", "in-kernel"):
            ld = 3 * D + (32 if form == "columns" else 0)
            qkv = torch.randn(planes, M, ld, device=DEV).to(dt)
            xa = torch.randn(planes, M, D, device=DEV).to(dt)
            if planes == 2:
                qkv[1] *= 2.0 ** -10; xa[1] *= 2.0 ** -10
            out = torch.empty(planes, M, D, dtype=dt, device=DEV)
            offs = torch.arange(0, M + 1, T, dtype=torch.int32, device=DEV)
            table = torch.randn(H, 2 * T - 1, device=DEV)
            cst = torch.randn(H, device=DEV)
            gst = torch.cat([torch.zeros(M, 1), torch.ones(M, 1)], 1).to(DEV).contiguous()
            gw, gcb = (torch.randn(planes, 2 * H, dh, device=DEV) * 0.05).to(dt), torch.zeros(H, 4, device=DEV)
            a = L.AttentionArgs()
            a.qkv, a.ld, a.plane_stride, a.q_col, a.k_col, a.v_col, a.B = qkv.data_ptr(), ld, M * ld, 0, D, 2 * D, B
            a.frame_offs, a.table, a.max_frames, a.table_T = offs.data_ptr(), table.data_ptr(), T, T
            a.out, a.ldo, a.out_plane_stride, a.H, a.dh, a.scale, a.mode = out.data_ptr(), D, M * D, H, dh, -1.0, mode
            a.gru_const, a.gate_col = cst.data_ptr(), 3 * D
            if form == "in-kernel":
                a.gate_x, a.gate_x_ld, a.gate_x_plane_stride, a.gate_x_planes = xa.data_ptr(), D, M * D, planes
                a.gate_stat, a.gate_w, a.gate_cb, a.gate_w_plane_stride = gst.data_ptr(), gw.data_ptr(), gcb.data_ptr(), 2 * H * dh
            run = lambda: L.check(L.lib.ser_attention_v(C.byref(a), st))
            for _ in range(3):
                run()
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            print(f"B={B:2d} T={T} {NAMES[mode]:6s} gate {form:9s}: {np.median(ts):7.1f} us", flush=True)
