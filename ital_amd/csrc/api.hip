// Error reporting shared by the entry points of libital_hip.so.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "ital_hip.h"
#include "ital_internal.h"

static thread_local char g_err[512] = "";

int ital_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int ital_check_launch(const char* who) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", who, hipGetErrorString(e));
    return -5;
}

extern "C" const char* ital_last_error(void) { return g_err; }
extern "C" const char* ital_version(void) { return "ital_hip 0.1 (gfx950)"; }
