// Version and thread-local error text of the C ABI.  Plain C++ (no HIP header): shared by libserhip.so and by the sanitizer
// build of the host-side file I/O (`make asan`, tests/test_host_fuzz.py).
#include "../../include/ser_hip.h"
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

extern "C" int ser_version(void) { return SER_ABI_VERSION; }
extern "C" const char* ser_last_error(void) { return g_err; }
