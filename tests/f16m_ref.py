"""Host restatement of the SER_MODE_FP16M operand format (include/ser_hip.h) for the kernel tests and tools -- test infrastructure, never
imported by the product.  fp32 matrix -> (fp16 hi plane, e4m3 cross-term bytes, E8M0 block-scale words) exactly as ser_pack_f16m and the GEMM /
row-kernel epilogues produce them, and the decode back to fp64.

Format (one row, one 64-column tile t):  bytes [128 t, 128 t + 64) = P, [128 t + 64, 128 t + 128) = Q of plane 1; activations P = x - hi, Q = x;
weights P = w, Q = w - hi; scale word (t, row) = codes [P cols 0-31, P cols 32-63, Q cols 0-31, Q cols 32-63], value 2^(code - 127), the
smallest power of two that brings the block's largest magnitude to <= 448."""
import numpy as np
import torch

F16_MAX = 65504.0


def _codes(amax: np.ndarray) -> np.ndarray:
    """mx_code of csrc/ser_common.h: ceil(log2(amax / 448)) + 127 from the fp32 bit pattern of amax * (1 / 448), clamped to [1, 254]"""
    t = (amax.astype(np.float32) * np.float32(1.0 / 448.0)).astype(np.float32)
    u = t.view(np.uint32).astype(np.uint64)
    c = (u + 0x7FFFFF) >> 23
    return np.clip(c, 1, 254).astype(np.uint32)


def _e4m3_bytes(v: np.ndarray) -> np.ndarray:
    """fp32 -> OCP e4m3fn bytes, round to nearest even (what v_cvt_pk_fp8_f32 does on gfx950 for |v| <= 464)"""
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def _e4m3_value(b: np.ndarray) -> np.ndarray:
    return torch.from_numpy(np.ascontiguousarray(b, dtype=np.uint8)).view(torch.float8_e4m3fn).to(torch.float64).numpy()


def pack(x: torch.Tensor, weight: bool):
    """fp32 CPU [R, C] (C % 64 == 0) -> dict(hi fp16 [R, C], x8 uint8 [R, 2 C] (plane 1 as bytes), scales uint32 [C / 64, R])"""
    x = x.detach().to(torch.float32).contiguous()
    R, Cn = x.shape
    assert Cn % 64 == 0
    hi = x.clamp(-F16_MAX, F16_MAX).to(torch.float16)
    xv = x.numpy()
    lo = (xv - hi.float().numpy()).astype(np.float32)

    def quant(t):
        blocks = t.reshape(R, Cn // 32, 32)
        code = _codes(np.abs(blocks).max(-1))                                   # [R, C / 32]
        inv = (np.uint32(254) - code).astype(np.uint32) << np.uint32(23)
        inv = inv.view(np.float32)
        return _e4m3_bytes(blocks * inv[:, :, None]).reshape(R, Cn), code

    bx, cx = quant(xv)
    bl, cl = quant(lo)
    P, Q, cp, cq = (bx, bl, cx, cl) if weight else (bl, bx, cl, cx)
    x8 = np.empty((R, Cn // 64, 128), dtype=np.uint8)
    x8[:, :, :64] = P.reshape(R, Cn // 64, 64)
    x8[:, :, 64:] = Q.reshape(R, Cn // 64, 64)
    cp, cq = cp.reshape(R, Cn // 64, 2), cq.reshape(R, Cn // 64, 2)
    words = cp[:, :, 0] | (cp[:, :, 1] << 8) | (cq[:, :, 0] << 16) | (cq[:, :, 1] << 24)
    return dict(hi=hi, x8=torch.from_numpy(x8.reshape(R, 2 * Cn)), scales=torch.from_numpy(words.T.astype(np.uint32).view(np.int32).copy()))


def decode(hi: torch.Tensor, x8: torch.Tensor, scales: torch.Tensor):
    """-> (hi, P, Q) as float64 [R, C]: what the matrix instructions multiply"""
    R, Cn = hi.shape
    b = x8.numpy().reshape(R, Cn // 64, 128)
    w = scales.numpy().view(np.uint32).T.reshape(R, Cn // 64)                  # [R, tiles]
    out = []
    for half, shifts in ((b[:, :, :64], (0, 8)), (b[:, :, 64:], (16, 24))):
        v = _e4m3_value(half).reshape(R, Cn // 64, 2, 32)
        for j, sh in enumerate(shifts):
            code = ((w >> np.uint32(sh)) & np.uint32(0xFF)).astype(np.float64)
            v[:, :, j, :] *= np.exp2(code - 127.0)[:, :, None]
        out.append(torch.from_numpy(v.reshape(R, Cn)))
    return hi.double(), out[0], out[1]


def product(a_planes, w_planes) -> torch.Tensor:
    """what ser_gemm(FP16M) accumulates, in float64: a_hi w_hi^T + P_a P_w^T + Q_a Q_w^T"""
    ah, ap, aq = a_planes
    wh, wp, wq = w_planes
    return ah @ wh.T + ap @ wp.T + aq @ wq.T


def to_device(p, device="cuda:0", extra_rows: int = 0):
    """packed dict -> (planes tensor fp16 [2, R + extra, C] with plane 1 holding the bytes, scales int32 [C / 64, R + extra]) on the device"""
    hi, x8, sc = p["hi"], p["x8"], p["scales"]
    R, Cn = hi.shape
    t = torch.zeros((2, R + extra_rows, Cn), dtype=torch.float16)
    t[0, :R] = hi
    t[1, :R] = x8.view(torch.float16).reshape(R, Cn)
    s = torch.zeros((Cn // 64, R + extra_rows), dtype=torch.int32)
    s[:, :R] = sc
    return t.to(device), s.to(device)
