// The one collective of a greedy step below the C ABI: every rank contributes its selection record,
// all ranks receive all of them -- ncclAllGather over RCCL (xGMI inside a node).  ~1.3 KB x world at d = 128 .. ~20 KB at
// d = 2048: latency bound, one ring step per peer.
//
// RCCL is resolved at run time from what the process ALREADY has loaded (a PyTorch host brings its own librccl.so; a
// second copy in the process would be handed communicators the first one created: undefined behaviour).  Order:
//   1. the global symbol scope (a host linked against RCCL, or one that loaded it RTLD_GLOBAL);
//   2. the loaded objects themselves (dl_iterate_phdr): a Python host loads torch's RCCL with RTLD_LOCAL, invisible to (1);
//      the object whose file name starts with "librccl" is re-opened by its own path with RTLD_NOLOAD (no new copy possible);
//   3. only if the environment names a library explicitly (ITAL_RCCL_LIBRARY=/path/librccl.so.1) is anything loaded by path.
// Nothing is ever loaded by bare name.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <link.h>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ital_hip.h"
#include "ital_internal.h"

namespace {

typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);   // ncclAllGather
typedef const char* (*error_string_fn)(int);
typedef int (*comm_int_fn)(void*, int*);                                              // ncclCommCount, ncclCommUserRank, ncclCommGetAsyncError
constexpr int NCCL_FLOAT64 = 8;   // ncclDouble (nccl.h: ncclFloat64 = 8)

all_gather_fn g_all_gather = nullptr;
error_string_fn g_error_string = nullptr;
comm_int_fn g_comm_count = nullptr, g_comm_rank = nullptr, g_comm_async_error = nullptr;
constexpr int NCCL_SUCCESS = 0, NCCL_IN_PROGRESS = 7;   // nccl.h: ncclSuccess, ncclInProgress (a non-blocking communicator still connecting)
char g_how[600] = "";

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
    const char* path = info->dlpi_name;
    if (!path || !*path) return 0;
    const char* base = strrchr(path, '/');
    base = base ? base + 1 : path;
    if (strncmp(base, "librccl", 7) != 0) return 0;
    snprintf(static_cast<char*>(out), 512, "%s", path);
    return 1;   // stop: first match
}

std::mutex g_resolve_lock;      // the entry points are resolved once, whichever thread asks first (a failed look-up may be
                                // repeated later: the host can load RCCL after this library)

bool resolve() {
    std::lock_guard<std::mutex> hold(g_resolve_lock);
    if (g_all_gather) return true;
    void* handle = nullptr;
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
    if (sym) {
        snprintf(g_how, sizeof(g_how), "global symbol scope");
    } else {
        char path[512] = "";
        if (dl_iterate_phdr(find_loaded_rccl, path) && (handle = dlopen(path, RTLD_NOW | RTLD_NOLOAD))) {
            sym = dlsym(handle, "ncclAllGather");
            snprintf(g_how, sizeof(g_how), "already loaded %s", path);
        }
    }
    if (!sym) {
        const char* env = getenv("ITAL_RCCL_LIBRARY");
        if (env && *env && (handle = dlopen(env, RTLD_NOW | RTLD_LOCAL))) {
            sym = dlsym(handle, "ncclAllGather");
            snprintf(g_how, sizeof(g_how), "ITAL_RCCL_LIBRARY=%s", env);
        }
    }
    if (!sym) return false;
    auto look = [&](const char* name) { return handle ? dlsym(handle, name) : dlsym(RTLD_DEFAULT, name); };
    g_error_string = reinterpret_cast<error_string_fn>(look("ncclGetErrorString"));
    g_comm_count = reinterpret_cast<comm_int_fn>(look("ncclCommCount"));
    g_comm_rank = reinterpret_cast<comm_int_fn>(look("ncclCommUserRank"));
    g_comm_async_error = reinterpret_cast<comm_int_fn>(look("ncclCommGetAsyncError"));
    g_all_gather = reinterpret_cast<all_gather_fn>(sym);      // last: what `resolve` tests first
    return true;
}

}  // namespace

extern "C" int ital_select_exchange(const double* record, double* records_all, int rec_len, void* nccl_comm,
                                    hipStream_t stream) {
    if (!record || !records_all || rec_len <= 0) return ital_fail(-22, "ital_select_exchange: bad arguments");
    if (!nccl_comm) return ital_fail(-22, "ital_select_exchange: communicator missing (ncclComm_t of this rank)");
    if (!resolve())
        return ital_fail(-38, "ital_select_exchange: no RCCL (ncclAllGather) loaded in this process -- create the communicator "
                              "with the RCCL the host uses first, or name the library in ITAL_RCCL_LIBRARY");
    const int rc = g_all_gather(record, records_all, (size_t)rec_len, NCCL_FLOAT64, nccl_comm, stream);
    if (rc != 0) {
        char msg[256];
        snprintf(msg, sizeof(msg), "ital_select_exchange: ncclAllGather failed: %s", g_error_string ? g_error_string(rc) : "?");
        return ital_fail(-5, msg);
    }
    return 0;
}

extern "C" int ital_exchange_info(void* nccl_comm, int* world, int* rank, char* how, int how_len) {
    if (!nccl_comm) return ital_fail(-22, "ital_exchange_info: communicator missing (ncclComm_t of this rank)");
    if (!resolve()) return ital_fail(-38, "ital_exchange_info: no RCCL (ncclAllGather) loaded in this process");
    if (how && how_len > 0) snprintf(how, (size_t)how_len, "%s", g_how);
    if (!g_comm_count || !g_comm_rank) return ital_fail(-38, "ital_exchange_info: the loaded RCCL lacks ncclCommCount / ncclCommUserRank");
    int w = -1, r = -1;
    const int rc = g_comm_count(nccl_comm, &w) ? -5 : (g_comm_rank(nccl_comm, &r) ? -5 : 0);
    if (rc) return ital_fail(-5, "ital_exchange_info: ncclCommCount / ncclCommUserRank failed");
    if (world) *world = w;
    if (rank) *rank = r;
    return 0;
}

// What RCCL's progress thread says about the communicator (ncclCommGetAsyncError): calls issued below torch.distributed are
// invisible to its watchdog, so the host polls this while it waits for a round's picks (ital_amd.ital: _await_picks).
extern "C" int ital_exchange_error(void* nccl_comm, int* async_error) {
    if (async_error) *async_error = -1;
    if (!nccl_comm) return ital_fail(-22, "ital_exchange_error: communicator missing (ncclComm_t of this rank)");
    if (!resolve()) return ital_fail(-38, "ital_exchange_error: no RCCL (ncclAllGather) loaded in this process");
    if (!g_comm_async_error) return ital_fail(-38, "ital_exchange_error: the loaded RCCL lacks ncclCommGetAsyncError");
    int err = -1;
    const int rc = g_comm_async_error(nccl_comm, &err);
    if (rc != NCCL_SUCCESS) {
        char msg[256];
        snprintf(msg, sizeof(msg), "ital_exchange_error: ncclCommGetAsyncError failed: %s", g_error_string ? g_error_string(rc) : "?");
        return ital_fail(-5, msg);
    }
    if (async_error) *async_error = err;
    if (err != NCCL_SUCCESS && err != NCCL_IN_PROGRESS) {
        char msg[256];
        snprintf(msg, sizeof(msg), "ital_exchange_error: the communicator reports an asynchronous error: %s",
                 g_error_string ? g_error_string(err) : "?");
        return ital_fail(-5, msg);
    }
    return 0;
}
