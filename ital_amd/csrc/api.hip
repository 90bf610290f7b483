// Error reporting shared by the entry points of libital_hip.so.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "ital_hip.h"
#include "ital_internal.h"

static thread_local char g_err[512] = "";
long long g_ital_launches = 0;

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

int ital_raise_lds_limit(const void* kernel, int bytes, ItalLdsFlags& flags, const char* who) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return ital_fail(-5, "cannot tell the current device");
    if (flags.done[dev]) return 0;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: cannot raise the dynamic LDS limit to %d bytes", who, bytes);
        (void)hipGetLastError();
        return -12;
    }
    flags.done[dev] = true;
    return 0;
}

extern "C" const char* ital_last_error(void) { return g_err; }
extern "C" int64_t ital_launch_count(void) { return g_ital_launches; }
extern "C" const char* ital_version(void) { return "ital_hip 0.1 (gfx950)"; }
