// One fetch_unlabelled(k) round of the perfect-user path as ONE call below the C ABI (reference ital/ital.py:98-134): candidate
// list upkeep, then per greedy step the scorer (ending with the selection inside its last launch; several ranks: with the
// rank's record, followed by the record exchange -- ncclAllGather -- and the resolve launch) and the new member's
// cross-covariance column.  Nothing here computes anything new -- it is ital_score_step / ital_cross_cov_cols enqueued in a
// C loop: at 9298 candidates the first two greedy steps are 10 - 25 us kernels and a Python host needs 10 - 17 us per launch
// (descriptor marshalling through ctypes), the GPU idles in between; from C the launches are ~3 us apart.
//
// Candidate list upkeep on the device: between two rounds of the retrieval loop the candidate list loses exactly the
// samples of the last batch (reference retrieval_base.py:78-87: ascending indices without the seen ones).  Instead of a new
// list from the host every round (build, pageable upload -- a synchronising copy -- and two fill launches) one single-
// workgroup launch compacts the previous list by its alive flags, re-arms the flags and clears the round's status slot.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ital_hip.h"
#include "ital_internal.h"
#include "qmc_seed.h"

// internal entry points of score.hip / rbf.hip
int ital_score_step_internal(const ital_score_desc* d, hipStream_t stream, bool seeds_ready, ital::SeedArgs* seeds_only);
int ital_cross_cov_cols_seed(const double* X, const double* xnorm, int64_t n, int ldx, const double* Xs, const double* sn, int c,
                             const double* W, int ldw, const double* V, int64_t ldv, int m, double var, double length_scale,
                             double* out, int64_t ldo, const ital::SeedArgs& seed, hipStream_t stream);

namespace ital {

// mode 0: alive[0 .. n) = 1, ret[kmax] = 0.   mode 1: additionally cand_new = cand_prev[alive_prev != 0] (order kept),
// n = number of survivors the host expects; a different count raises status bit 8.
__global__ __launch_bounds__(1024) void fetch_begin_kernel(int mode, const int32_t* __restrict__ cand_prev, int64_t n_prev,
                                                           int32_t* __restrict__ cand_new, uint8_t* __restrict__ alive,
                                                           int64_t n, int64_t* __restrict__ ret, int kmax,
                                                           int* __restrict__ status) {
    __shared__ int wave_tot[16];
    __shared__ int s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (mode == 1) {
        const int64_t chunk = (n_prev + blockDim.x - 1) / blockDim.x;
        const int64_t lo = (int64_t)tid * chunk, hi = lo + chunk < n_prev ? lo + chunk : n_prev;
        int cnt = 0;
        for (int64_t p = lo; p < hi; p++) cnt += alive[p] != 0;
        // exclusive prefix of the per-thread counts: wave scan, then the wave totals
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; w++) base += wave_tot[w];
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); w++) tot += wave_tot[w];
            s_total = tot;
        }
        int64_t at = base + incl - cnt;
        for (int64_t p = lo; p < hi; p++)
            if (alive[p] != 0) cand_new[at++] = cand_prev[p];
        __syncthreads();                      // all flags read before any is re-armed
        if (tid == 0 && s_total != n) atomicOr(status, 8);
    }
    for (int64_t p = tid; p < n; p += blockDim.x) alive[p] = 1;
    if (tid == 0) ret[kmax] = 0;
}

// The same upkeep for long lists (a million candidates on one rank: the single workgroup above walks them in 4.5 ms) as three
// launches over tiles of 16 384 entries -- 16 consecutive flags per thread: (1) live entries per tile; (2) every tile
// compacts itself behind the tiles in front of it (their counts: at most 128 numbers); (3) once all flags are read: re-arm,
// clear the status slot, compare the survivors with what the host expects.
constexpr int64_t COMPACT_TILE = 16384;
constexpr int64_t COMPACT_FROM = 1 << 16;      // shorter lists: the single-workgroup kernel (one launch)

__device__ __forceinline__ unsigned tile_flags(const uint8_t* __restrict__ alive, int64_t at, int64_t n_prev) {
    unsigned m = 0;                            // bit j: entry at + j is alive
    if (at + 16 <= n_prev) {
        const uint64_t* w = reinterpret_cast<const uint64_t*>(alive + at);      // at is a multiple of 16, the buffer 256-byte aligned
        const uint64_t lo = w[0], hi = w[1];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            m |= (((lo >> (8 * j)) & 0xffull) != 0 ? 1u : 0u) << j;
            m |= (((hi >> (8 * j)) & 0xffull) != 0 ? 1u : 0u) << (8 + j);
        }
    } else {
        for (int j = 0; j < 16 && at + j < n_prev; j++) m |= (alive[at + j] != 0 ? 1u : 0u) << j;
    }
    return m;
}

__device__ __forceinline__ int block_sum_1024(int v, int* s_wave) {   // sum over the workgroup, valid in every thread
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) s_wave[wave] = v;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < 16; w++) tot += s_wave[w];
    __syncthreads();
    return tot;
}

__global__ __launch_bounds__(1024) void compact_count_kernel(const uint8_t* __restrict__ alive, int64_t n_prev, int* __restrict__ counts) {
    __shared__ int s_wave[16];
    const int64_t at = (int64_t)blockIdx.x * COMPACT_TILE + (int64_t)threadIdx.x * 16;
    const int cnt = at < n_prev ? __popc(tile_flags(alive, at, n_prev)) : 0;
    const int tot = block_sum_1024(cnt, s_wave);
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void compact_scatter_kernel(const int32_t* __restrict__ cand_prev, const uint8_t* __restrict__ alive,
                                                               int64_t n_prev, int32_t* __restrict__ cand_new,
                                                               const int* __restrict__ counts) {
    __shared__ int s_wave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t base = 0;
    for (int b = 0; b < (int)blockIdx.x; b++) base += counts[b];
    const int64_t at = (int64_t)blockIdx.x * COMPACT_TILE + (int64_t)threadIdx.x * 16;
    const unsigned m = at < n_prev ? tile_flags(alive, at, n_prev) : 0u;
    const int cnt = __popc(m);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = incl - cnt;
    for (int w = 0; w < wave; w++) before += s_wave[w];
    int64_t out = base + before;
    for (unsigned mm = m; mm; mm &= mm - 1) cand_new[out++] = cand_prev[at + __builtin_ctz(mm)];
}

__global__ __launch_bounds__(1024) void compact_rearm_kernel(uint8_t* __restrict__ alive, int64_t n, int64_t* __restrict__ ret, int kmax,
                                                             int* __restrict__ status, const int* __restrict__ counts, int ntiles) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) alive[p] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t tot = 0;
        for (int b = 0; b < ntiles; b++) tot += counts[b];
        if (tot != n) atomicOr(status, 8);
        ret[kmax] = 0;
    }
}

}  // namespace ital

using namespace ital;

extern "C" int ital_fetch_round(const ital_round_desc* r, hipStream_t stream) {
    if (!r) return ital_fail(-22, "ital_fetch_round: null descriptor");
    const ital_score_desc& tpl = r->step;
    if (r->k < 1 || r->k > ITAL_MAX_T || r->k > tpl.batch.kmax) return ital_fail(-22, "ital_fetch_round: k outside 1..min(ITAL_MAX_T, kmax)");
    const bool ranks = r->world > 0;
    if (ranks && (!r->records_all || (!r->nccl_comm && !r->exchange)))
        return ital_fail(-22, "ital_fetch_round: several ranks need records_all and a transport (nccl_comm or exchange)");
    if (ranks ? tpl.n_cand < 1 : tpl.n_cand < r->k)
        return ital_fail(-22, "ital_fetch_round: fewer candidates than greedy steps (several ranks: none on this rank)");
    if (tpl.n_cand > ITAL_ROUND_MAX_CAND) return ital_fail(-22, "ital_fetch_round: more than ITAL_ROUND_MAX_CAND candidates (use the per-step entry points)");
    if (!tpl.sel_record || !tpl.sel_ret) return ital_fail(-22, "ital_fetch_round: the round selects inside the scorer: sel_* missing");
    if (r->begin < 0 || r->begin > 2) return ital_fail(-22, "ital_fetch_round: begin must be 0, 1 or 2");
    if (r->begin == 2 && (!r->cand_prev || r->n_prev < tpl.n_cand || r->n_prev > ITAL_ROUND_MAX_CAND || r->cand_prev == tpl.cand))
        return ital_fail(-22, "ital_fetch_round: begin = 2 needs the previous list in a buffer of its own");
    if (r->begin == 2 && r->n_prev >= COMPACT_FROM) {
        // one count per tile (<= ITAL_ROUND_MAX_CAND / 16 384 + 1 = 129 ints) in the CALLER's memory: the head of step.sel_parts,
        // which holds nothing between two rounds (every scoring launch writes its partials before the selecting block reads
        // them) -- two learners on different streams of one device share no scratch of the library (round-4 advice)
        const unsigned ntiles = (unsigned)((r->n_prev + COMPACT_TILE - 1) / COMPACT_TILE);
        if (!tpl.sel_parts || tpl.sel_parts_len * 2 < (int64_t)ntiles + 1)
            return ital_fail(-22, "ital_fetch_round: step.sel_parts too small for the tile counts of the list upkeep (ital_sel_parts_len)");
        int* counts = reinterpret_cast<int*>(tpl.sel_parts);
        uint8_t* alive = const_cast<uint8_t*>(tpl.alive);
        ITAL_LAUNCH(compact_count_kernel, dim3(ntiles), dim3(1024), 0, stream, alive, r->n_prev, counts);
        ITAL_LAUNCH(compact_scatter_kernel, dim3(ntiles), dim3(1024), 0, stream, r->cand_prev, alive, r->n_prev,
                    const_cast<int32_t*>(tpl.cand), counts);
        ITAL_LAUNCH(compact_rearm_kernel, dim3(ntiles), dim3(1024), 0, stream, alive, tpl.n_cand, tpl.sel_ret, tpl.batch.kmax,
                    tpl.status, counts, (int)ntiles);
        const int rc = ital_check_launch("ital_fetch_round(begin, tiles)");
        if (rc) return rc;
    } else if (r->begin) {
        ITAL_LAUNCH(fetch_begin_kernel, dim3(1), dim3(1024), 0, stream, r->begin - 1, r->cand_prev, r->n_prev,
                    const_cast<int32_t*>(tpl.cand), const_cast<uint8_t*>(tpl.alive), tpl.n_cand, tpl.sel_ret, tpl.batch.kmax,
                    tpl.status);
        const int rc = ital_check_launch("ital_fetch_round(begin)");
        if (rc) return rc;
    }
    auto step_desc = [&](int t) {
        ital_score_desc d = tpl;
        d.t = t;
        if (r->mi_keep) d.mi = r->mi_keep + (int64_t)(t - 1) * tpl.n_cand;
        if (t >= 3) {
            d.jump = r->jump[t];
            d.jumppat = r->jumppat[t];
            d.vk = r->vk[t];
            for (int j = 0; j < 6; j++) d.seed[j] = r->seeds[t][j];
        }
        d.ev_start = r->ev_start[t];
        d.ev_stop = r->ev_stop[t];
        if (ranks) d.sel_ret = nullptr;      // the scoring launch stops after this rank's record
        return d;
    };
    const int rec_len = ITAL_REC_HEADER + tpl.sel_ldx + tpl.sel_ldw + tpl.batch.kmax;
    bool seeds_ready = false;
    for (int t = 1; t <= r->k; t++) {
        const ital_score_desc d = step_desc(t);
        int rc = ital_score_step_internal(&d, stream, seeds_ready, nullptr);
        if (rc) return rc;
        seeds_ready = false;
        if (ranks) {
            rc = r->exchange ? r->exchange(r->exchange_ctx, tpl.sel_record, r->records_all, rec_len, stream)
                             : ital_select_exchange(tpl.sel_record, r->records_all, rec_len, r->nccl_comm, stream);
            if (rc) return r->exchange ? ital_fail(rc, "ital_fetch_round: the host's exchange failed") : rc;
            rc = ital_select_resolve(r->records_all, r->world, rec_len, tpl.sel_rank, 0, t - 1, tpl.batch,
                                     const_cast<uint8_t*>(tpl.alive), tpl.sel_ret, stream);
            if (rc) return rc;
        }
        if (t < r->k) {
            const int slot = t - 1;
            const double* xb = tpl.batch.XB + (int64_t)slot * tpl.batch.ldx;
            const double* vb = tpl.batch.VB + (int64_t)slot * tpl.batch.ldw;
            double* col = const_cast<double*>(tpl.C) + (int64_t)slot * tpl.ldc;
            // the generator states of step t + 1's candidates ride along with the column (both depend on the selection just
            // made and on nothing else) when that step runs in one slab of its workspace
            SeedArgs seed;
            const ital_score_desc nx = step_desc(t + 1);
            if (ital_score_step_internal(&nx, stream, false, &seed) == 0) {
                rc = ital_cross_cov_cols_seed(tpl.sel_X, tpl.sel_xnorm, r->n_rows, tpl.sel_ldx, xb, tpl.batch.XBn + slot, 1, vb,
                                              tpl.batch.ldw, tpl.sel_V, tpl.sel_ldv, tpl.sel_m, r->var, r->length_scale, col,
                                              tpl.ldc, seed, stream);
                seeds_ready = true;
            } else {
                rc = ital_cross_cov_cols(tpl.sel_X, tpl.sel_xnorm, r->n_rows, tpl.sel_ldx, xb, tpl.batch.XBn + slot, 1, vb,
                                         tpl.batch.ldw, tpl.sel_V, tpl.sel_ldv, tpl.sel_m, r->var, r->length_scale, col, tpl.ldc,
                                         stream);
            }
            if (rc) return rc;
        }
    }
    return 0;
}

extern "C" int ital_mcmi_round(const ital_mcmi_round_desc* r, hipStream_t stream) {
    if (!r) return ital_fail(-22, "ital_mcmi_round: null descriptor");
    const ital_mcmi_desc& tpl = r->step;
    if (r->k < 1 || r->k > ITAL_MAX_T || r->k > tpl.batch.kmax) return ital_fail(-22, "ital_mcmi_round: k outside 1..min(ITAL_MAX_T, kmax)");
    if (tpl.pos_offset != 0 || tpl.n_i != tpl.n_all) return ital_fail(-22, "ital_mcmi_round: one rank scores the whole block (pos_offset 0, n_i == n_all)");
    if (tpl.n_all < r->k) return ital_fail(-22, "ital_mcmi_round: fewer candidates than greedy steps");
    if (tpl.n_all > (1 << 18)) return ital_fail(-22, "ital_mcmi_round: more than 2^18 candidates (use the per-step entry points)");
    if (!r->Xc || !r->xnc || !r->pos || !r->record || !r->ret || !tpl.alive || !tpl.cov || (r->m > 0 && !r->Vc))
        return ital_fail(-22, "ital_mcmi_round: null argument");
    uint8_t* alive = const_cast<uint8_t*>(tpl.alive);
    int rc;
    if (r->begin) {
        ITAL_LAUNCH(fetch_begin_kernel, dim3(1), dim3(1024), 0, stream, 0, nullptr, (int64_t)0, nullptr, alive, tpl.n_i, r->ret,
                    tpl.batch.kmax, const_cast<int*>(r->status));
        if ((rc = ital_check_launch("ital_mcmi_round(begin)"))) return rc;
    }
    rc = ital_cov_block(r->Xc, r->xnc, tpl.n_i, r->Xc, r->xnc, tpl.n_all, r->ldx, r->Vc, r->ldv, r->Vc, r->ldv, r->m, r->var,
                        r->length_scale, const_cast<double*>(tpl.cov), tpl.ld_cov, stream);
    if (rc) return rc;
    for (int t = 1; t <= r->k; t++) {
        ital_mcmi_desc d = tpl;
        d.t = t;
        if ((rc = ital_mcmi_score_step(&d, stream))) return rc;
        rc = ital_select_fused(tpl.ce, r->pos, alive, tpl.n_i, 0, nullptr, 0, 0, 1, tpl.mu, tpl.s2, r->Xc, r->xnc, r->ldx, r->Vc,
                               r->ldv, r->m, r->ldw, tpl.C, tpl.ldc, t - 1, t - 1, tpl.batch, r->status, r->record, r->ret, stream);
        if (rc) return rc;
        if (t < r->k) {
            const int slot = t - 1;
            rc = ital_cross_cov_cols(r->Xc, r->xnc, tpl.n_all, r->ldx, tpl.batch.XB + (int64_t)slot * tpl.batch.ldx,
                                     tpl.batch.XBn + slot, 1, tpl.batch.VB + (int64_t)slot * tpl.batch.ldw, tpl.batch.ldw, r->Vc,
                                     r->ldv, r->m, r->var, r->length_scale, const_cast<double*>(tpl.C) + (int64_t)slot * tpl.ldc,
                                     tpl.ldc, stream);
            if (rc) return rc;
        }
    }
    return 0;
}
