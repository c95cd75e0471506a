// C-ABI plumbing of libserhip: launch check, workspace sizing, command lists (version + error text: hosterr.hip).
#include "ser_common.h"
#include <stdarg.h>
#include <stdio.h>

int ser_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        return ser_fail((int)e, "%s: launch failed: %s", what, hipGetErrorString(e));
    }
    return 0;
}


extern "C" size_t ser_workspace_bytes(int op, int B, int T, int D, int H, int mode) {
    (void)T; (void)D; (void)H; (void)mode;
    switch (op) {
        case SER_WS_LOGMEL:
            // per-block partial maxima [B][256] floats + the fp64 twiddle table [400][208] x (cos, -sin) + the Hann window (ser_logmel_init)
            return (size_t)(B > 0 ? B : 1) * 1024 + (size_t)400 * 208 * 16 + 400 * 4;
        case SER_WS_WAVE_FRAMES:
            return (size_t)(B > 0 ? B : 1) * 64 * 2 * sizeof(double);      // [B][64] partial (sum, sum^2)
        default:
            return 0;
    }
}


// ---- command lists (see ser_hip.h): one foreign call per batch instead of one per kernel ------------------------
extern "C" int ser_run(const ser_cmd* cmds, int32_t n, int32_t* failed_at, void* stream) {
    if (!cmds || n < 0) return ser_fail(-1, "ser_run: bad command list");
    for (int32_t i = 0; i < n; ++i) {
        const ser_cmd& c = cmds[i];
        int rc;
        switch (c.op) {
            case SER_OP_GEMM:
                rc = ser_gemm(&c.u.gemm, stream);
                break;
            case SER_OP_ATTENTION: {
                rc = ser_attention_v(&c.u.attention, stream);
                break;
            }
            case SER_OP_LAYERNORM: {
                rc = ser_layernorm_v(&c.u.layernorm, stream);
                break;
            }
            case SER_OP_WAVE_FRAMES: {
                rc = ser_wave_frames_v(&c.u.wave_frames, stream);
                break;
            }
            case SER_OP_ROW_CENTER: {
                rc = ser_row_center_v(&c.u.row_center, stream);
                break;
            }
            case SER_OP_LOGMEL: {
                const ser_logmel_args& a = c.u.logmel;
                rc = ser_logmel_whisper(a.wav, a.sample_offs, a.B, a.mel, a.n_mels, a.out, a.work, stream);
                break;
            }
            case SER_OP_PACK_ACT: {
                rc = ser_pack_act_v(&c.u.pack_act, stream);
                break;
            }
            default:
                rc = ser_fail(-2, "ser_run: command %d has unknown op %d", i, c.op);
        }
        if (rc != 0) {
            if (failed_at) *failed_at = i;
            return rc;
        }
    }
    return 0;
}
