// Context-style entry points of the C ABI (ital_ctx_create / _fit / _update / _fetch / ...): a learner
// whose device buffers the LIBRARY owns, for hosts that do not want to manage the ~20 buffers of the descriptor API
// themselves (tests/host_gpu_driver.cpp is that host; ital_amd's own Python learners stay on the descriptor API, whose
// buffers are torch tensors).  Everything here is host code over the descriptor entry points of this same library: the
// perfect-user path of ITAL -- fit, update (Cholesky append + whitening sweep), fetch_unlabelled(k <= 8) with full
// sign-pattern enumeration, predict_stored -- on one rank, or on several with one ncclAllGather of a record per greedy step
// (ital_select_local -> ital_select_exchange -> ital_select_resolve).
//
// Reference: ital/retrieval_base.py:34-61 (fit / reset), :105-126 (update), ital/ital.py:84-134 (fetch_unlabelled),
// ital/gp.py:141-232 (fit / update / predict_stored).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "ital_hip.h"
#include "ital_internal.h"

struct ital_ctx {
    int64_t n_total = 0, row0 = 0, row1 = 0, n = 0, ldv = 0;
    int d = 0, ldx = 0, cap = 0, kmax = ITAL_MAX_T, rank = 0, world = 1, m = 0;
    double length_scale = 1, var = 1, noise = 1e-6;
    void* comm = nullptr;
    bool fitted = false;
    // device memory (all owned here)
    double *X = nullptr, *xn = nullptr, *L = nullptr, *alpha = nullptr, *XT = nullptr, *XTn = nullptr, *V = nullptr, *mu = nullptr,
           *s2 = nullptr, *ybuf = nullptr, *C = nullptr, *mi = nullptr, *rec = nullptr, *rec_all = nullptr, *work3k = nullptr,
           *qwork = nullptr, *stage = nullptr;
    int64_t qwork_doubles = 0, cand_cap = 0;
    int* status = nullptr;
    int32_t* cand = nullptr;
    uint8_t* alive = nullptr;
    int64_t* ret = nullptr;
    ital_batch batch = {};
    long long* jump[ITAL_MAX_T + 1] = {};
    long long* jumppat[ITAL_MAX_T + 1] = {};
    double* vk[ITAL_MAX_T + 1] = {};
    // host bookkeeping
    int mvn_state[6] = {};
    std::vector<uint8_t> seen;          // [n_total] labelled (the reference's relevant / irrelevant ids)
    int64_t n_seen = 0;
    std::vector<int64_t> last_picks;    // the batch of the last fetch (its feature rows sit in batch.XB on every rank)
    std::vector<void*> owned;
};

namespace {

template <class T>
T* dalloc(ital_ctx* c, size_t count) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T) + 64) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T) + 64) != hipSuccess) {
        (void)hipFree(p);
        return nullptr;
    }
    c->owned.push_back(p);
    return static_cast<T*>(p);
}

// Gives a buffer of the context back (its replacement has been allocated): nothing may still be using it.
void dfree(ital_ctx* c, void* p) {
    if (!p) return;
    const auto it = std::find(c->owned.begin(), c->owned.end(), p);
    if (it != c->owned.end()) c->owned.erase(it);
    (void)hipFree(p);
}

int pad16(int64_t v) { return (int)((v + 15) / 16 * 16); }

}  // namespace

extern "C" int ital_ctx_destroy(ital_ctx* c) {
    if (!c) return 0;
    (void)hipDeviceSynchronize();
    for (void* p : c->owned) (void)hipFree(p);
    delete c;
    return 0;
}

extern "C" int ital_ctx_create(int64_t n_total, int d, double length_scale, double var, double noise, int capacity, int rank,
                               int world, void* nccl_comm, ital_ctx** out) {
    if (!out) return ital_fail(-22, "ital_ctx_create: out missing");
    *out = nullptr;
    if (n_total < 1 || d < 1 || world < 1 || rank < 0 || rank >= world || !(length_scale > 0) || !(var > 0) || !(noise >= 0))
        return ital_fail(-22, "ital_ctx_create: bad arguments");
    if (world > 1 && !nccl_comm) return ital_fail(-22, "ital_ctx_create: several ranks need this rank's ncclComm_t");
    ital_ctx* c = new ital_ctx();
    c->n_total = n_total;
    c->row0 = n_total * rank / world;                      // contiguous row blocks (ital_amd.sharding.row_range)
    c->row1 = n_total * (rank + 1) / world;
    c->n = c->row1 - c->row0;
    c->d = d;
    c->ldx = pad16(d);
    c->ldv = pad16(std::max<int64_t>(c->n, 1));
    c->cap = pad16(capacity > 0 ? capacity : 256);
    c->length_scale = length_scale; c->var = var; c->noise = noise;
    c->rank = rank; c->world = world; c->comm = nccl_comm;
    const int kmax = c->kmax, ldx = c->ldx, cap = c->cap;
    const int rec_len = ital_record_len(ldx, cap, kmax);
    c->X = dalloc<double>(c, (size_t)std::max<int64_t>(c->n, 1) * ldx);
    c->xn = dalloc<double>(c, std::max<int64_t>(c->n, 1));
    c->L = dalloc<double>(c, (size_t)cap * cap);
    c->alpha = dalloc<double>(c, cap);
    c->XT = dalloc<double>(c, (size_t)cap * ldx);
    c->XTn = dalloc<double>(c, cap);
    c->V = dalloc<double>(c, (size_t)cap * c->ldv);
    c->mu = dalloc<double>(c, c->ldv);
    c->s2 = dalloc<double>(c, c->ldv);
    c->ybuf = dalloc<double>(c, 16);
    c->status = dalloc<int>(c, 1);
    c->C = dalloc<double>(c, (size_t)kmax * c->ldv);
    c->ret = dalloc<int64_t>(c, kmax + 1);
    c->rec = dalloc<double>(c, rec_len);
    c->rec_all = dalloc<double>(c, (size_t)world * rec_len);
    c->work3k = dalloc<double>(c, 3 * 1024);
    c->batch.kmax = kmax; c->batch.ldx = ldx; c->batch.ldw = cap;
    c->batch.bidx = dalloc<int64_t>(c, kmax);
    c->batch.bgpos = dalloc<int64_t>(c, kmax);
    c->batch.bsort = dalloc<int32_t>(c, kmax);
    c->batch.bmu = dalloc<double>(c, kmax);
    c->batch.sig = dalloc<double>(c, (size_t)kmax * kmax);
    c->batch.XB = dalloc<double>(c, (size_t)kmax * ldx);
    c->batch.XBn = dalloc<double>(c, kmax);
    c->batch.VB = dalloc<double>(c, (size_t)kmax * cap);
    // every allocation (an unchecked failure would surface as a device fault in the first kernel that touches the buffer)
    const void* all[] = {c->X, c->xn, c->L, c->alpha, c->XT, c->XTn, c->V, c->mu, c->s2, c->ybuf, c->status, c->C, c->ret, c->rec,
                         c->rec_all, c->work3k, c->batch.bidx, c->batch.bgpos, c->batch.bsort, c->batch.bmu, c->batch.sig,
                         c->batch.XB, c->batch.XBn, c->batch.VB};
    for (const void* p : all)
        if (!p) {
            ital_ctx_destroy(c);
            return ital_fail(-12, "ital_ctx_create: out of device memory");
        }
    c->seen.assign((size_t)n_total, 0);
    ital_mvn_seed(c->mvn_state);
    *out = c;
    return 0;
}

// This rank's rows [row0, row1) of the n_total x d matrix, row-major with leading dimension d; host memory, or device memory
// when on_device != 0.  Resets the labelled set (reference retrieval_base.py:34-61).
extern "C" int ital_ctx_fit(ital_ctx* c, const double* rows, int on_device, hipStream_t stream) {
    if (!c || (!rows && c->n > 0)) return ital_fail(-22, "ital_ctx_fit: bad arguments");
    if (c->n > 0 &&
        hipMemcpy2DAsync(c->X, (size_t)c->ldx * sizeof(double), rows, (size_t)c->d * sizeof(double), (size_t)c->d * sizeof(double),
                         (size_t)c->n, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream) != hipSuccess)
        return ital_fail(-5, "ital_ctx_fit: copy of the rows failed");
    if (c->n > 0) {
        const int rc = ital_row_norms(c->X, c->n, c->ldx, c->xn, stream);
        if (rc) return rc;
    }
    // prior: mean 0, variance var; no labelled sample
    if (hipMemsetAsync(c->mu, 0, (size_t)c->ldv * sizeof(double), stream) != hipSuccess) return ital_fail(-5, "ital_ctx_fit: memset failed");
    std::vector<double> v((size_t)c->ldv, c->var);
    if (hipMemcpyAsync(c->s2, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return ital_fail(-5, "ital_ctx_fit: upload failed");
    (void)hipMemsetAsync(c->status, 0, sizeof(int), stream);
    c->m = 0;
    std::fill(c->seen.begin(), c->seen.end(), 0);
    c->n_seen = 0;
    c->last_picks.clear();
    c->fitted = true;
    return 0;
}

// Labels c_new samples (global indices idx, labels y = +-1): rank-c Cholesky append + whitened rows + refresh of the means and
// variances (reference retrieval_base.py:105-126, gp.py:164-200).  The feature rows come from this rank's own rows, or --
// several ranks -- from the replicated batch state when the samples are (a subset of) the batch just fetched, which is the
// retrieval loop; anything else on several ranks is -38.
extern "C" int ital_ctx_update(ital_ctx* c, const int64_t* idx, const double* y, int c_new, hipStream_t stream) {
    if (!c || !c->fitted || !idx || !y || c_new < 1) return ital_fail(-22, "ital_ctx_update: bad arguments");
    if (c->m + c_new > c->cap) return ital_fail(-12, "ital_ctx_update: labelled-set capacity exceeded (ital_ctx_create: capacity)");
    for (int j = 0; j < c_new; j++) {
        if (idx[j] < 0 || idx[j] >= c->n_total) return ital_fail(-22, "ital_ctx_update: index outside the data");
        if (c->seen[(size_t)idx[j]]) return ital_fail(-22, "ital_ctx_update: Cannot change feedback once given.");
    }
    bool all_batch = !c->last_picks.empty(), all_local = true;
    std::vector<int> slot(c_new);
    for (int j = 0; j < c_new; j++) {
        const auto it = std::find(c->last_picks.begin(), c->last_picks.end(), idx[j]);
        if (it == c->last_picks.end()) all_batch = false;
        else slot[j] = (int)(it - c->last_picks.begin());
        if (idx[j] < c->row0 || idx[j] >= c->row1) all_local = false;
    }
    // Samples outside the batch just fetched (the first labels of a session: a query) on a communicator: their feature rows
    // are replicated through the exchange the greedy steps use -- one all-gather of a record-sized buffer per sample, the
    // owner contributes the row, everybody keeps the owner's part (the reference slices the rows out of the matrix every
    // worker holds, gp.py:185-190).
    const bool via_comm = !all_batch && c->comm != nullptr;
    if (!all_batch && !all_local && !via_comm)
        return ital_fail(-38, "ital_ctx_update: rows of other ranks need the communicator (ital_ctx_create: nccl_comm)");
    const int rec_len = ital_record_len(c->ldx, c->cap, c->kmax);
    for (int j0 = 0; j0 < c_new; j0 += 16) {
        const int cc = std::min(16, c_new - j0);
        ital_label_batch lb;
        memset(&lb, 0, sizeof(lb));
        lb.c = cc;
        for (int j = 0; j < cc; j++) {
            lb.slot[j] = all_batch ? slot[j0 + j] : (via_comm ? j : (int)(idx[j0 + j] - c->row0));
            lb.y[j] = y[j0 + j];
        }
        if (via_comm) {
            if (!c->stage) {
                c->stage = dalloc<double>(c, (size_t)16 * c->ldx);
                if (!c->stage) return ital_fail(-12, "ital_ctx_update: out of device memory");
            }
            for (int j = 0; j < cc; j++) {
                const int64_t gi = idx[j0 + j];
                int owner = 0;
                while (owner + 1 < c->world && c->n_total * (owner + 1) / c->world <= gi) owner++;      // rows [n r / w, n (r + 1) / w)
                if (hipMemsetAsync(c->rec, 0, (size_t)rec_len * sizeof(double), stream) != hipSuccess) return ital_fail(-5, "ital_ctx_update: memset failed");
                if (owner == c->rank &&
                    hipMemcpyAsync(c->rec, c->X + (size_t)(gi - c->row0) * c->ldx, (size_t)c->ldx * sizeof(double), hipMemcpyDeviceToDevice,
                                   stream) != hipSuccess)
                    return ital_fail(-5, "ital_ctx_update: copy of the row failed");
                const int rc_x = ital_select_exchange(c->rec, c->rec_all, rec_len, c->comm, stream);
                if (rc_x) return rc_x;
                if (hipMemcpyAsync(c->stage + (size_t)j * c->ldx, c->rec_all + (size_t)owner * rec_len, (size_t)c->ldx * sizeof(double),
                                   hipMemcpyDeviceToDevice, stream) != hipSuccess)
                    return ital_fail(-5, "ital_ctx_update: copy of the replicated row failed");
            }
        }
        const int m = c->m;
        int rc = ital_stage_labelled(all_batch ? c->batch.XB : (via_comm ? c->stage : c->X), c->ldx, lb, c->XT + (size_t)m * c->ldx, c->XTn + m, c->ybuf, stream);
        if (!rc) rc = ital_chol_append(c->XT, c->XTn, c->ldx, c->L, c->cap, c->alpha, c->ybuf, m, cc, c->var, c->length_scale, c->noise,
                                       c->status, stream);
        if (!rc) rc = ital_whiten_append(c->X, c->xn, c->n, c->ldx, c->XT + (size_t)m * c->ldx, c->XTn + m, cc, c->L + (size_t)m * c->cap,
                                         c->cap, c->L + (size_t)m * c->cap + m, c->alpha + m, c->V, c->ldv, m, c->var, c->length_scale,
                                         c->mu, c->s2, stream);
        if (rc) return rc;
        c->m += cc;
    }
    for (int j = 0; j < c_new; j++) c->seen[(size_t)idx[j]] = 1;
    c->n_seen += c_new;
    c->last_picks.clear();
    return 0;
}

// fetch_unlabelled(k): the k picks in selection order into picks[0 .. k) (host memory).  Perfect user, full enumeration of
// the 2^t sign patterns (k <= ITAL_MAX_T), candidates = all unlabelled samples in ascending order
// (reference ital.py:84-134, retrieval_base.py:78-87).  Synchronises `stream` (the picks are its result).
// -71: the round met a batch the fast scorer does not cover (duplicate samples, large noise: status bits 2 / 4) -- such a
// round belongs to ital_score_generic, which this convenience layer does not drive.
extern "C" int ital_ctx_fetch(ital_ctx* c, int k, int64_t* picks, hipStream_t stream) {
    if (!c || !c->fitted || !picks) return ital_fail(-22, "ital_ctx_fetch: bad arguments");
    if (c->m == 0) return ital_fail(-22, "ital_ctx_fetch: needs a fitted relevance model: call ital_ctx_update first");
    const int64_t n_unseen = c->n_total - c->n_seen;
    if (k > n_unseen) k = (int)n_unseen;
    if (k < 1) return 0;
    if (k > ITAL_MAX_T) return ital_fail(-22, "ital_ctx_fetch: batches larger than ITAL_MAX_T need the Monte-Carlo switch (ital_score_generic)");
    // candidate list: ascending unseen samples; this rank's share and the list position of its first entry
    std::vector<int32_t> cand_h;
    int64_t pos_offset = 0;
    for (int64_t i = 0; i < c->row0; i++) pos_offset += c->seen[(size_t)i] ? 0 : 1;
    for (int64_t i = c->row0; i < c->row1; i++)
        if (!c->seen[(size_t)i]) cand_h.push_back((int32_t)(i - c->row0));
    const int64_t nc = (int64_t)cand_h.size();
    if (nc > c->cand_cap) {
        // (sized for all of this rank's rows at once: grows at most once per context; nothing enqueued reads the old ones --
        // every fetch ends with a stream synchronisation)
        dfree(c, c->cand); dfree(c, c->alive); dfree(c, c->mi);
        c->cand_cap = std::max<int64_t>(nc, c->n);
        c->cand = dalloc<int32_t>(c, c->cand_cap);
        c->alive = dalloc<uint8_t>(c, c->cand_cap);
        c->mi = dalloc<double>(c, c->cand_cap);
        if (!c->cand || !c->alive || !c->mi) {
            c->cand_cap = 0;
            return ital_fail(-12, "ital_ctx_fetch: out of device memory");
        }
    }
    // the lattice scorer's workspace, ONCE for the whole round: what its largest step (t = k) needs, capped at 1 GiB (slabs
    // beyond) -- ital_round_workspace is documented for exactly this.  (Until round 6 it was sized per step: each of the steps
    // t = 3 .. k allocated a larger buffer and left the previous one in `owned` until ital_ctx_destroy, up to ~5 GiB stranded
    // after the first k = 8 fetch at large n.)
    if (k >= 3) {
        const int64_t want = ital_round_workspace(k, std::max<int64_t>(nc, 1), (int64_t)1 << 27);
        if (want > c->qwork_doubles) {
            if (hipStreamSynchronize(stream) != hipSuccess) return ital_fail(-5, "ital_ctx_fetch: stream error");
            dfree(c, c->qwork);
            c->qwork = dalloc<double>(c, (size_t)want);
            c->qwork_doubles = c->qwork ? want : 0;
            if (!c->qwork) return ital_fail(-12, "ital_ctx_fetch: out of device memory (workspace)");
        }
    }
    if (nc > 0 && (hipMemcpyAsync(c->cand, cand_h.data(), nc * sizeof(int32_t), hipMemcpyHostToDevice, stream) != hipSuccess ||
                   hipMemsetAsync(c->alive, 1, nc, stream) != hipSuccess))
        return ital_fail(-5, "ital_ctx_fetch: upload of the candidate list failed");
    (void)hipMemsetAsync(c->ret, 0, (c->kmax + 1) * sizeof(int64_t), stream);
    if (hipStreamSynchronize(stream) != hipSuccess) return ital_fail(-5, "ital_ctx_fetch: stream error");     // (cand_h leaves scope)
    const int rec_len = ital_record_len(c->ldx, c->cap, c->kmax);
    int64_t n_alive = n_unseen;
    for (int t = 1; t <= k; t++) {
        ital_score_desc desc;
        memset(&desc, 0, sizeof(desc));
        desc.t = t; desc.n_cand = nc; desc.cand = c->cand; desc.alive = c->alive; desc.mu = c->mu; desc.s2 = c->s2;
        desc.C = c->C; desc.ldc = c->ldv; desc.row_offset = c->row0; desc.pos_offset = pos_offset; desc.batch = c->batch;
        desc.noise = c->noise; desc.eps = 1e-12; desc.mi = c->mi; desc.status = c->status;
        if (t >= 3) {
            if (!c->jump[t]) {
                std::vector<long long> jump((size_t)ITAL_JUMP_BITS * 18), pat((size_t)(1 << t) * 18);
                std::vector<double> vk(t - 1);
                int rc = ital_mvn_tables(t, jump.data(), pat.data(), vk.data());
                if (rc) return rc;
                c->jump[t] = dalloc<long long>(c, jump.size());
                c->jumppat[t] = dalloc<long long>(c, pat.size());
                c->vk[t] = dalloc<double>(c, vk.size());
                if (!c->jump[t] || !c->jumppat[t] || !c->vk[t]) return ital_fail(-12, "ital_ctx_fetch: out of device memory");
                if (hipMemcpy(c->jump[t], jump.data(), jump.size() * sizeof(long long), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(c->jumppat[t], pat.data(), pat.size() * sizeof(long long), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(c->vk[t], vk.data(), vk.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
                    return ital_fail(-5, "ital_ctx_fetch: upload of the stream tables failed");
            }
            desc.jump = c->jump[t]; desc.jumppat = c->jumppat[t]; desc.vk = c->vk[t];
            desc.work = c->qwork; desc.work_doubles = c->qwork_doubles;
            for (int j = 0; j < 6; j++) desc.seed[j] = c->mvn_state[j];
        }
        int rc = nc > 0 ? ital_score_step(&desc, stream) : 0;
        if (rc) return rc;
        if (c->world == 1 && !c->comm) {
            rc = ital_select_fused(c->mi, c->cand, c->alive, nc, pos_offset, nullptr, c->row0, c->rank, 0, c->mu, c->s2, c->X, c->xn,
                                   c->ldx, c->V, c->ldv, c->m, c->cap, c->C, c->ldv, t - 1, t - 1, c->batch, c->status, c->rec, c->ret,
                                   stream);
        } else {
            rc = ital_select_local(c->mi, c->cand, c->alive, nc, pos_offset, nullptr, c->row0, c->rank, 0, c->mu, c->s2, c->X, c->xn,
                                   c->ldx, c->V, c->ldv, c->m, c->cap, c->C, c->ldv, t - 1, c->kmax, c->status, c->work3k, c->rec,
                                   stream);
            if (!rc) rc = ital_select_exchange(c->rec, c->rec_all, rec_len, c->comm, stream);
            if (!rc) rc = ital_select_resolve(c->rec_all, c->world, rec_len, c->rank, 0, t - 1, c->batch, c->alive, c->ret, stream);
        }
        if (rc) return rc;
        if (t < k) {
            const int slot = t - 1;
            rc = ital_cross_cov_cols(c->X, c->xn, c->n, c->ldx, c->batch.XB + (size_t)slot * c->ldx, c->batch.XBn + slot, 1,
                                     c->batch.VB + (size_t)slot * c->cap, c->cap, c->V, c->ldv, c->m, c->var, c->length_scale,
                                     c->C + (size_t)slot * c->ldv, c->ldv, stream);
            if (rc) return rc;
        }
        // the serial reference has now made 2 * 2^t mvndst calls per live candidate of the WHOLE list (ital.py:191-206)
        ital_mvn_advance(c->mvn_state, n_alive * (int64_t)(2 << t) * ital_mvn_draws_per_call(t));
        n_alive--;
    }
    std::vector<int64_t> host((size_t)c->kmax + 1);
    if (hipMemcpyAsync(host.data(), c->ret, host.size() * sizeof(int64_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return ital_fail(-5, "ital_ctx_fetch: download of the picks failed");
    const int64_t st = host[(size_t)c->kmax];
    if (st & 1) return ital_fail(-33, "ital_ctx_fetch: kernel matrix of the labelled samples is not positive definite");
    if (st & 6) {
        int zero = 0;
        (void)hipMemcpy(c->status, &zero, sizeof(int), hipMemcpyHostToDevice);
        return ital_fail(-71, "ital_ctx_fetch: the round needs the general scorer (duplicate samples in the batch or large noise)");
    }
    c->last_picks.assign(host.begin(), host.begin() + k);
    for (int t = 0; t < k; t++) picks[t] = host[(size_t)t];
    return k;
}

// Predictive mean and variance of this rank's rows (n_local doubles each, host memory; either may be NULL); the variance
// clamped at 0 as predict_stored(cov_mode='diag') does (reference gp.py:203-232).  Synchronises `stream`.
extern "C" int ital_ctx_predict_stored(ital_ctx* c, double* mean, double* variance, hipStream_t stream) {
    if (!c || !c->fitted) return ital_fail(-22, "ital_ctx_predict_stored: bad arguments");
    if (mean && hipMemcpyAsync(mean, c->mu, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess)
        return ital_fail(-5, "ital_ctx_predict_stored: download failed");
    if (variance && hipMemcpyAsync(variance, c->s2, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess)
        return ital_fail(-5, "ital_ctx_predict_stored: download failed");
    if (hipStreamSynchronize(stream) != hipSuccess) return ital_fail(-5, "ital_ctx_predict_stored: stream error");
    if (variance)
        for (int64_t i = 0; i < c->n; i++) variance[i] = variance[i] > 0 ? variance[i] : 0;
    return 0;
}

extern "C" int64_t ital_ctx_local_rows(const ital_ctx* c, int64_t* row0) {
    if (!c) return 0;
    if (row0) *row0 = c->row0;
    return c->n;
}
