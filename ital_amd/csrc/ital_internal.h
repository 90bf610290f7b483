// Internal helpers shared by the HIP translation units of libital_hip.so.
#pragma once
#include <hip/hip_runtime.h>

// Records `msg` as the library's last error and returns `code` (negative errno-style).
int ital_fail(int code, const char* msg);
// hipGetLastError() after a launch; 0 or a recorded failure.
int ital_check_launch(const char* who);

// Raises a kernel's dynamic-LDS limit once per device (the attribute belongs to the function on ONE device; a process that
// drives several GPUs must set it on each).  `done`: the caller's static per-device flags.  0 or a recorded failure.
struct ItalLdsFlags { bool done[16] = {}; };
int ital_raise_lds_limit(const void* kernel, int bytes, ItalLdsFlags& flags, const char* who);

// Every kernel launch of the library goes through this: counts the launches (ital_launch_count(): launches per round are
// part of what bench.py reports -- short greedy steps are launch-latency bound).
extern long long g_ital_launches;
#define ITAL_LAUNCH(...) do { ++g_ital_launches; hipLaunchKernelGGL(__VA_ARGS__); } while (0)
