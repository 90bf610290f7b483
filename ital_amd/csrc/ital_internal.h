// Internal helpers shared by the HIP translation units of libital_hip.so.
#pragma once
#include <hip/hip_runtime.h>

// Records `msg` as the library's last error and returns `code` (negative errno-style).
int ital_fail(int code, const char* msg);
// hipGetLastError() after a launch; 0 or a recorded failure.
int ital_check_launch(const char* who);
