// The one collective of a greedy step below the C ABI: every rank contributes its selection record,
// all ranks receive all of them -- ncclAllGather over RCCL (xGMI inside a node).  ~1.3 KB x world at d = 128 .. ~20 KB at
// d = 2048: latency bound, one ring step per peer.
//
// RCCL is resolved at run time from what the process already has loaded (a PyTorch host brings its own librccl.so;
// linking a second copy into this library would give the process two RCCL instances) and only then by name.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "ital_hip.h"
#include "ital_internal.h"

namespace {

typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);   // ncclAllGather
typedef const char* (*error_string_fn)(int);
constexpr int NCCL_FLOAT64 = 8;   // ncclDouble (nccl.h: ncclFloat64 = 8)

all_gather_fn g_all_gather = nullptr;
error_string_fn g_error_string = nullptr;

bool resolve() {
    if (g_all_gather) return true;
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
    if (!sym) {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h && (sym = dlsym(h, "ncclAllGather"))) break;
        }
    }
    if (!sym) return false;
    g_all_gather = reinterpret_cast<all_gather_fn>(sym);
    g_error_string = reinterpret_cast<error_string_fn>(dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
    return true;
}

}  // namespace

extern "C" int ital_select_exchange(const double* record, double* records_all, int rec_len, void* nccl_comm,
                                    hipStream_t stream) {
    if (!record || !records_all || rec_len <= 0) return ital_fail(-22, "ital_select_exchange: bad arguments");
    if (!nccl_comm) return ital_fail(-22, "ital_select_exchange: communicator missing (ncclComm_t of this rank)");
    if (!resolve()) return ital_fail(-38, "ital_select_exchange: RCCL (ncclAllGather) not found in this process");
    const int rc = g_all_gather(record, records_all, (size_t)rec_len, NCCL_FLOAT64, nccl_comm, stream);
    if (rc != 0) {
        char msg[256];
        snprintf(msg, sizeof(msg), "ital_select_exchange: ncclAllGather failed: %s", g_error_string ? g_error_string(rc) : "?");
        return ital_fail(-5, msg);
    }
    return 0;
}
