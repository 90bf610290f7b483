// Arg-extreme over the scored candidates and the batch bookkeeping of a greedy step (role R9/R10).
//
// np.argmax / np.argmin semantics of the reference (ital/ital.py:130, ital/mcmi.py:77): the FIRST extreme in
// candidate-list order wins, and a NaN beats every number (first NaN wins).  Each rank reduces its own live
// positions and packs a fixed-size record (score, list position, data index, mean, variance, the feature row,
// the whitened column and the cross-covariances with the members so far); after the records of all ranks have
// been gathered, every rank runs the same resolve step and appends the winner to the replicated batch state.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "ital_hip.h"
#include "ital_internal.h"
#include "select_common.h"

namespace ital {

__global__ __launch_bounds__(256) void select_partial_kernel(const double* __restrict__ mi, const uint8_t* __restrict__ alive,
                                                             int64_t n_cand, int64_t pos_offset,
                                                             const int64_t* __restrict__ gpos, int mode,
                                                             double* __restrict__ work) {
    Best v = scan_best(mi, alive, n_cand, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x,
                       pos_offset, gpos, mode);
    v = block_best(v, mode);
    if (threadIdx.x == 0) {
        work[3 * blockIdx.x] = v.val;
        work[3 * blockIdx.x + 1] = (double)v.pos;
        work[3 * blockIdx.x + 2] = (double)v.loc;
    }
}

__global__ __launch_bounds__(256) void select_record_kernel(RecordArgs a) {
    Best v = {0.0, -1, 0};
    for (int i = threadIdx.x; i < a.nparts; i += blockDim.x) {
        Best c = {a.work[3 * i], (int64_t)a.work[3 * i + 1], (int64_t)a.work[3 * i + 2]};
        if (better(c, v, a.mode)) v = c;
    }
    v = block_best(v, a.mode);
    record_body(a, v);
}

// Local arg-extreme and record in one single-workgroup launch (small problems: the two-stage reduction above costs a
// launch more than it saves).
__global__ __launch_bounds__(1024) void select_local_small_kernel(const double* __restrict__ mi, const uint8_t* __restrict__ alive,
                                                                  int64_t n_cand, RecordArgs a) {
    Best v = scan_best(mi, alive, n_cand, threadIdx.x, blockDim.x, a.pos_offset, a.gpos, a.mode);
    v = block_best(v, a.mode);
    record_body(a, v);
}

__global__ __launch_bounds__(256) void select_resolve_kernel(const double* __restrict__ records, int world, int rec_len,
                                                             int rank, int mode, int slot, ital_batch b,
                                                             uint8_t* __restrict__ alive, int64_t* __restrict__ ret) {
    resolve_body(records, world, rec_len, rank, mode, slot, b, alive, ret);
}

// One rank: arg-extreme over all positions, record and resolve in ONE single-workgroup launch (three launches otherwise;
// the greedy steps of small problems are launch-latency bound).
__global__ __launch_bounds__(1024) void select_fused_kernel(const double* __restrict__ mi, int64_t n_cand, RecordArgs a,
                                                            int slot, ital_batch b, uint8_t* alive, int64_t* __restrict__ ret) {
    Best v = scan_best(mi, alive, n_cand, threadIdx.x, blockDim.x, a.pos_offset, a.gpos, a.mode);
    v = block_best(v, a.mode);
    record_body(a, v);
    __threadfence_block();
    __syncthreads();
    const int rec_len = ITAL_REC_HEADER + a.ldx + a.ldw + a.kmax;
    resolve_body(a.record, 1, rec_len, a.rank, a.mode, slot, b, alive, ret);
}

}  // namespace ital

using namespace ital;

extern "C" int ital_select_local(const double* mi, const int32_t* cand, const uint8_t* alive, int64_t n_cand,
                                 int64_t pos_offset, const int64_t* gpos, int64_t row_offset, int rank, int mode,
                                 const double* mu, const double* s2, const double* X, const double* xnorm, int ldx,
                                 const double* V, int64_t ldv, int m, int ldw, const double* C, int64_t ldc, int nprev,
                                 int kmax, const int* status, double* work, double* record, hipStream_t stream) {
    if (mode != 0 && mode != 1) return ital_fail(-22, "ital_select_local: mode must be 0 (argmax) or 1 (argmin)");
    if (m > ldw) return ital_fail(-22, "ital_select_local: m exceeds ldw");
    if (n_cand <= (1 << 18)) {
        RecordArgs a1 = {cand, pos_offset, gpos, row_offset, rank, mode, 0, mu, s2, X, xnorm, ldx, V, ldv, m, ldw,
                         C, ldc, nprev, kmax, nullptr, record, status};
        ITAL_LAUNCH(select_local_small_kernel, dim3(1), dim3(1024), 0, stream, mi, alive, n_cand, a1);
        return ital_check_launch("ital_select_local(small)");
    }
    int nparts = (int)((n_cand + 255) / 256);
    if (nparts > 1024) nparts = 1024;
    if (nparts < 1) nparts = 1;
    ITAL_LAUNCH(select_partial_kernel, dim3(nparts), dim3(256), 0, stream, mi, alive, n_cand, pos_offset, gpos, mode,
                       work);
    int rc = ital_check_launch("ital_select_local(partial)");
    if (rc) return rc;
    RecordArgs a = {cand, pos_offset, gpos, row_offset, rank, mode, nparts, mu, s2, X, xnorm, ldx, V, ldv, m, ldw,
                    C, ldc, nprev, kmax, work, record, status};
    ITAL_LAUNCH(select_record_kernel, dim3(1), dim3(256), 0, stream, a);
    return ital_check_launch("ital_select_local(record)");
}

extern "C" int ital_select_fused(const double* mi, const int32_t* cand, uint8_t* alive, int64_t n_cand, int64_t pos_offset,
                                 const int64_t* gpos, int64_t row_offset, int rank, int mode, const double* mu,
                                 const double* s2, const double* X, const double* xnorm, int ldx, const double* V, int64_t ldv,
                                 int m, int ldw, const double* C, int64_t ldc, int nprev, int slot, ital_batch batch,
                                 const int* status, double* record, int64_t* ret, hipStream_t stream) {
    if (mode != 0 && mode != 1) return ital_fail(-22, "ital_select_fused: mode must be 0 (argmax) or 1 (argmin)");
    if (m > ldw) return ital_fail(-22, "ital_select_fused: m exceeds ldw");
    if (slot < 0 || slot >= batch.kmax) return ital_fail(-22, "ital_select_fused: slot outside the batch capacity");
    if (ldx != batch.ldx || ldw != batch.ldw) return ital_fail(-22, "ital_select_fused: batch layout mismatch");
    RecordArgs a = {cand, pos_offset, gpos, row_offset, rank, mode, 0, mu, s2, X, xnorm, ldx, V, ldv, m, ldw,
                    C, ldc, nprev, batch.kmax, nullptr, record, status};
    ITAL_LAUNCH(select_fused_kernel, dim3(1), dim3(1024), 0, stream, mi, n_cand, a, slot, batch, alive, ret);
    return ital_check_launch("ital_select_fused");
}

extern "C" int ital_select_resolve(const double* records, int world, int rec_len, int rank, int mode, int slot,
                                   ital_batch batch, uint8_t* alive, int64_t* ret, hipStream_t stream) {
    if (slot < 0 || slot >= batch.kmax) return ital_fail(-22, "ital_select_resolve: slot outside the batch capacity");
    if (rec_len != ITAL_REC_HEADER + batch.ldx + batch.ldw + batch.kmax)
        return ital_fail(-22, "ital_select_resolve: record length does not match the batch layout");
    ITAL_LAUNCH(select_resolve_kernel, dim3(1), dim3(256), 0, stream, records, world, rec_len, rank, mode, slot,
                       batch, alive, ret);
    return ital_check_launch("ital_select_resolve");
}
