// C-ABI plumbing of libserhip: version, thread-local error text, workspace sizing.
#include "ser_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int ser_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int ser_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int ser_version(void) { return SER_ABI_VERSION; }
extern "C" const char* ser_last_error(void) { return g_err; }

extern "C" size_t ser_workspace_bytes(int op, int B, int T, int D, int H, int mode) {
    (void)T; (void)D; (void)H; (void)mode;
    switch (op) {
        case SER_WS_LOGMEL:
            // one 256-byte granule per utterance (running max) + the fp64 twiddle table [400][201]x2
            return (size_t)(B > 0 ? B : 1) * 256 + (size_t)400 * 201 * 16;
        case SER_WS_WAVE_FRAMES:
            return (size_t)(B > 0 ? B : 1) * 64 * 2 * sizeof(double);      // [B][64] partial (sum, sum^2)
        default:
            return 0;
    }
}
