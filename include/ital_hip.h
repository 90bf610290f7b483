/*
 * libital_hip.so -- C ABI of the MI355X (gfx950) hot path of ITAL's GP-based mutual-information
 * candidate selection.  Plain pointers and sizes only: every pointer is a *borrowed device pointer*
 * (the caller owns the memory, e.g. through torch tensors), every call is asynchronous on the given
 * HIP stream, returns 0 or a negative errno-style code (message: ital_last_error()).  No allocation
 * crosses the ABI (workspaces are the caller's), nothing synchronises the device, so a whole fetch_unlabelled(k) round can be
 * enqueued (or graph-captured) without host round trips.
 *
 * Threading: ONE caller thread per process, as the reference's learners (module-global state, ital/ital.py:619-621).  The
 * library keeps a little process-wide state without locks -- the last error message, the launch counter
 * (ital_launch_count: instrumentation), the RCCL entry points ital_select_exchange / ital_exchange_info resolve on first
 * use, the internal streams / scratch of ital_score_generic and ital_fetch_round per device.  A few small device
 * allocations of that kind are made by the library itself on first use (per device, never freed; a graph capture must
 * therefore follow one uncaptured call).
 *
 * The reference (cvjena/ITAL) is pure Python and has no FFI; each entry point below replaces the native
 * numerical routine the reference reaches at the cited place.  INTEGRATION.md shows the ctypes binding
 * a maintainer of the reference would add.
 */
#ifndef ITAL_HIP_H
#define ITAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t; /* same declaration as hip_runtime_api.h */

#define ITAL_MAX_T 8        /* largest batch dimension with full sign-pattern enumeration on the device */
#define ITAL_REC_HEADER 10  /* doubles in front of the feature row inside a selection record */
#define ITAL_JUMP_BITS 48
#define ITAL_GENERIC_MAX_DIM 20  /* largest orthant dimension of the general scorer (subset + picks + candidate) */
#define ITAL_GENERIC_MAX_REL 16  /* largest number of enumerated / sampled variables of the general scorer */
#define ITAL_GENERIC_MAX_CALLS (1 << 22) /* orthant probabilities per candidate and greedy step */
#define ITAL_TOPK_MAX 4096  /* largest k of ital_topk */
#define ITAL_ROUND_MAX_CAND (1 << 21) /* most candidates per rank of ital_fetch_round (one million rows on one GPU fit) */

/* Library identification / error reporting. */
const char* ital_version(void);
const char* ital_last_error(void);
/* Kernels this library has launched in the calling process so far (instrumentation: launches per round). */
int64_t ital_launch_count(void);

/* ---- GP core ------------------------------------------------------------------------------------------
 * Feature rows are fp64, row-major, leading dimension ldx = d rounded up to a multiple of 16 (zero padded).
 */

/* |x_i|^2 for every row.  Replaces np.sum(a**2, axis=-1), reference ital/gp.py:411,414-415. */
int ital_row_norms(const double* X, int64_t n, int ldx, double* xnorm, hipStream_t stream);

/* out[j][i] = var*exp(-|xs_j - x_i|^2 / (2 l^2)) for c <= 16 selected rows against all n rows.
 * Replaces GaussianProcess.kernel (np.dot + numexpr exp), reference ital/gp.py:390-416, for the columns
 * the streaming formulation needs (the reference forms the full N x N matrix at gp.py:128). */
int ital_rbf_cols(const double* X, const double* xnorm, int64_t n, int ldx, const double* Xs, const double* sn,
                  int c, double var, double length_scale, double* out, int64_t ldo, hipStream_t stream);

/* out[j][i] = k(xs_j, x_i) - sum_r W[j][r] V[r][i]: posterior cross-covariance of c points with all rows
 * (W[j] = whitened column of point j).  Replaces predict_cov_batch's cov_base_test, reference ital/gp.py:250-256. */
int ital_cross_cov_cols(const double* X, const double* xnorm, int64_t n, int ldx, const double* Xs,
                        const double* sn, int c, const double* W, int ldw, const double* V, int64_t ldv, int m,
                        double var, double length_scale, double* out, int64_t ldo, hipStream_t stream);

/* Rank-c Cholesky append of K[T,T] + noise*I (rows m..m+c-1 of L, alpha = L^-1 y).  XT rows m..m+c-1 must
 * hold the new feature rows, XTn their squared norms.  status |= 1 when the matrix is not positive definite.
 * Replaces invh / GaussianProcess.fit / update (dpotrf + dpotri from scratch), reference ital/gp.py:8-37,
 * :141-161, :164-200. */
int ital_chol_append(const double* XT, const double* XTn, int ldx, double* L, int ldl, double* alpha,
                     const double* ynew, int m, int c, double var, double length_scale, double noise, int* status,
                     hipStream_t stream);

/* Stages the c <= 16 samples an update labels: XT_dst[j][:] = rows[slot[j]][:] (feature rows, e.g. out of the batch state
 * XB of the last fetch), XTn_dst[j] = their squared norms, y_dst[j] = y[j]; the picks and labels travel by value.
 * Replaces the row slicing of GaussianProcess.update, reference ital/gp.py:185-190 (K_all[ind] / X[ind]). */
typedef struct ital_label_batch {
    int c;
    int slot[16];
    double y[16];
} ital_label_batch;
int ital_stage_labelled(const double* rows, int ldx, ital_label_batch lb, double* XT_dst, double* XTn_dst, double* y_dst,
                        hipStream_t stream);

/* ital_stage_labelled + ital_chol_append + ital_whiten_append of one update() as ONE call (two launches: the rows are staged
 * by the workgroup that appends to the factor).  rows[lb.slot[j]] = feature row of new sample j, lb.y[j] its label; XT, XTn,
 * L, alpha, V, mu, s2 as in the three entry points; ybuf: 16 doubles of scratch.
 * Replaces GaussianProcess.update + the predict_stored refresh, reference ital/gp.py:164-200, ital/retrieval_base.py:119-120. */
typedef struct ital_append_desc {
    const double* rows;
    int ldx;
    ital_label_batch lb;
    const double* X;
    const double* xnorm;
    int64_t n;
    double* XT;
    double* XTn;
    double* L;
    int ldl;
    double* alpha;
    double* ybuf;
    double* V;
    int64_t ldv;
    int m;
    double var, length_scale, noise;
    double* mu;
    double* s2;
    int* status;
} ital_append_desc;
int ital_gp_append(const ital_append_desc* a, hipStream_t stream);

/* Appends the c whitened rows V[m..m+c-1][:] = L22^-1 (K[new,:] - L21 V) and refreshes the predictive mean
 * mu += V_new^T alpha_new and variance s2 -= colsum(V_new^2) of every row.  L21 = &L[m][0] (ld ldw),
 * L22 = &L[m][m] (ld ldw).  Replaces predict_stored after an update, reference ital/gp.py:203-232 as called from
 * ital/retrieval_base.py:120 and ital/ital.py:558. */
int ital_whiten_append(const double* X, const double* xnorm, int64_t n, int ldx, const double* Xnew,
                       const double* snew, int c, const double* L21, int ldw, const double* L22,
                       const double* alpha_new, double* V, int64_t ldv, int m, double var, double length_scale,
                       double* mu, double* s2, hipStream_t stream);

/* Predictive mean and variance at nt external points: the points are whitened against the labelled set in sweeps of 16
 * labelled points (the MFMA kernel of ital_whiten_append), mean = Vt^T alpha, pvar = var - colsum(Vt^2) (clamped at 0
 * when `clamp`).  xtn: nt doubles of scratch (squared norms); Vt: [m][ldvt] receives the whitened columns of the test
 * points -- the full predictive covariance is then ital_cov_block(Xt, xtn, nt, Xt, xtn, nt, ldx, Vt, ldvt, Vt, ldvt, m, ...).
 * Replaces GaussianProcess.predict, reference ital/gp.py:264-292 (cov_mode None / 'diag' / 'full'). */
int ital_predict(const double* Xt, int64_t nt, int ldx, const double* XT, const double* XTn, int m, const double* L,
                 int ldl, const double* alpha, double var, double length_scale, double* mean, double* pvar, double* xtn,
                 double* Vt, int64_t ldvt, int clamp, hipStream_t stream);

/* The k largest of v[0..n) in descending order (NaN first, equal values by descending index -- the order of
 * np.argsort(v)[::-1][:k] with a stable sort): out_vals[k], out_idx[k] = index_offset + position.  Exact radix select on
 * the device, nothing of size n leaves it.  work: ital_topk_workspace() bytes of device memory.  1 <= k <= ITAL_TOPK_MAX,
 * k <= n.  Replaces ActiveRetrievalBase.top_results, reference ital/retrieval_base.py:64-75. */
int ital_topk(const double* v, int64_t n, int64_t index_offset, int k, double* out_vals, int64_t* out_idx, void* work,
              hipStream_t stream);
int64_t ital_topk_workspace(void);

/* ---- host-side bookkeeping of the reference's random streams (HOST pointers, no device work, usable without a GPU) ---
 *
 * SciPy's mvndst (reference ital/ital.py:380, :405, :425) draws its lattice shifts from MVNUNI, a generator whose state is
 * process-global and cannot be seeded.  The scorers replay that stream in the reference's serial evaluation order; the
 * host tells them where the stream stands.  A host that drives ital_score_step / ital_score_generic itself keeps one
 * state per process:
 *     int state[6]; ital_mvn_seed(state);                    once (the generator's DATA seed)
 *     ... desc.seed = state; ital_score_step(&desc, s); ...   per greedy step t
 *     ital_mvn_advance(state, n_alive * (2 << t) * ital_mvn_draws_per_call(t));   perfect user: 2 calls per sign pattern
 * (n_alive = live candidates of the whole list, all ranks; ital_score_generic: the draws_out / draws_in it was given).
 */
int ital_mvn_seed(int state[6]);
/* Uniforms one mvndst call with n finite-limit variables consumes: 8 (2 (n - 1) - 1), 0 for the closed forms n <= 2. */
int ital_mvn_draws_per_call(int n);
/* state <- state after n_draws more uniforms (3 x 3 matrix powers mod m1 / m2: O(log n_draws)). */
int ital_mvn_advance(int state[6], int64_t n_draws);
/* A whole round of k greedy steps with the perfect user (2 calls per sign pattern and live candidate): seeds[t] = the state
 * before step t (t = 1 .. k; what ital_round_desc.seeds takes), state <- the state after the round (n_cand live
 * candidates at step 1, one fewer per step). */
int ital_mvn_round_seeds(int state[6], int64_t n_cand, int k, int seeds[][6]);
/* Tables of ital_score_desc for batch dimension t (3 .. ITAL_MAX_T), to be copied to device memory by the caller; any of
 * the three may be NULL.  jump: [ITAL_JUMP_BITS][18], jumppat: [2^t][18], vk: [t - 1]. */
int ital_mvn_tables(int t, long long* jump, long long* jumppat, double* vk);
/* Tables of ital_gscore_desc: jump1 [ITAL_JUMP_BITS][18] (2^b uniforms), vk_all [nmax + 1][nmax] (row n: generators of
 * dimension n), nmax <= ITAL_GENERIC_MAX_DIM; either may be NULL. */
int ital_mvn_generic_tables(int nmax, long long* jump1, double* vk_all);

/* numpy's legacy global generator (MT19937 + polar method), the stream the reference's Monte-Carlo pattern sampler draws
 * its standard normals from: scipy.stats.multivariate_normal.rvs at reference ital/ital.py:297, one (t mc) x t block per
 * live candidate in list order.  The generator cannot jump (data-dependent rejections), but a rank that scores only its
 * own candidates need not COMPUTE the other ranks' normals: this advances the state past n_skip standard normals (raw
 * draws and accept tests only, ~4 ns each against ~40 ns), then writes the next n_out to out[] -- bit-identical to
 * np.random.standard_normal.  threads > 1: the values of a large request are produced by that many host threads (one more
 * skipping pass finds the generator state at the start of every thread's share).  State as
 * np.random.get_state(legacy=True) returns it / np.random.set_state takes it. */
typedef struct ital_np_legacy_state {
    uint32_t key[624];
    int32_t pos;         /* 0 .. 624 */
    int32_t has_gauss;
    double gauss;        /* cached second value of the last pair when has_gauss */
} ital_np_legacy_state;
int ital_np_legacy_normals(ital_np_legacy_state* st, int64_t n_skip, double* out, int64_t n_out, int threads);

/* ---- greedy batch construction ------------------------------------------------------------------------
 * Batch state, replicated on every rank, updated only by ital_select_resolve (device side, no host sync). */
typedef struct ital_batch {
    int kmax;          /* capacity (max batch size) */
    int ldx;           /* padded feature dimension */
    int ldw;           /* leading dimension of VB (>= labelled-set capacity) */
    int64_t* bidx;     /* [kmax] data index of each member, selection order */
    int64_t* bgpos;    /* [kmax] candidate-list position each member was picked from */
    int32_t* bsort;    /* [kmax] member slots ordered by data index */
    double* bmu;       /* [kmax] predictive means */
    double* sig;       /* [kmax][kmax] posterior covariance among the members */
    double* XB;        /* [kmax][ldx] feature rows */
    double* XBn;       /* [kmax] squared norms */
    double* VB;        /* [kmax][ldw] whitened columns */
} ital_batch;

typedef struct ital_score_desc {
    int t;                  /* batch dimension of this greedy step: members so far + 1 */
    int64_t n_cand;         /* candidate-list positions held by this rank */
    const int32_t* cand;    /* [n_cand] local row of each position (list order) */
    const uint8_t* alive;   /* [n_cand] 0 once picked */
    const double* mu;       /* [rows] predictive mean (rel_mean) */
    const double* s2;       /* [rows] predictive variance, NOT clamped */
    const double* C;        /* [t-1][ldc] cross-covariance of member b with every row */
    int64_t ldc;
    int64_t row_offset;     /* data index of local row 0 */
    int64_t pos_offset;     /* list position of local position 0 (used when gpos == NULL) */
    const int64_t* gpos;    /* [n_cand] list position of every local position when this rank's positions are not one
                               contiguous run of the list (top_candidates on several ranks, reference ital/ital.py:111-117) */
    ital_batch batch;
    double noise, eps;
    int label_mode;         /* 0 'mean', 1 'optimistic', 2 'pessimistic' (reference ital/ital.py:210-219) */
    double* mi;             /* [n_cand] out */
    /* t >= 3: replay of SciPy mvndst's MVNUNI stream (serial evaluation order of the reference) */
    int seed[6];            /* generator state before the first call of this greedy step */
    const long long* jump;  /* [ITAL_JUMP_BITS][18] transition matrices for 2^b calls of dimension t */
    const long long* jumppat; /* [2^t][18] transition matrices for 2r calls, r = 0..2^t-1 (the prior-probability call of
                               sign pattern r is call 2r of a candidate, the one after the simulated update call 2r+1) */
    const double* vk;       /* [t-1] Korobov generator vector of this dimension */
    int* status;            /* |= 2: singular conditional covariance met; |= 4: a probability after the simulated update is
                               not decided by its limits alone (noise > ~3e-4).  Either way mi[] is not valid and the step
                               belongs to ital_score_generic */
    /* t >= 3: workspace of `work_doubles` doubles in device memory, at least ital_score_workspace(t, 1); the candidates
     * are processed in slabs of work_doubles / ital_score_workspace(t, 1) (prepared calls: factor, limits, the 8 shifted
     * lattices of every evaluated call -- 0.5 to 1 KB per call) */
    double* work;
    int64_t work_doubles;
    void* ev_start;         /* optional hipEvent_t pair recorded on the stream right before the first / after the last */
    void* ev_stop;          /* lattice-sum kernel of the step (the FP64-VALU bound one).  One slab: that kernel alone; with
                               ceil(n_cand / slab) > 1 slabs the pair also spans the preparation / combine launches of the
                               slabs in between (a few percent of the lattice sums) */
    /* Optional, one rank: end the step with its selection -- everything ital_select_fused(mi, ..., nprev = slot = t - 1)
     * does -- inside the last scoring launch instead of a launch of its own (every block leaves its best candidate in
     * sel_parts, the block that finishes last selects, packs the record and appends the winner to the batch state; the
     * winner's alive flag is cleared, hence `alive` is written in this mode).  Off when sel_record == NULL. */
    const double* sel_X;    /* [rows][sel_ldx] feature rows, sel_xnorm their squared norms */
    const double* sel_xnorm;
    int sel_ldx;
    const double* sel_V;    /* [m][sel_ldv] whitened block */
    int64_t sel_ldv;
    int sel_m, sel_ldw;
    int sel_rank;
    double* sel_record;     /* scratch: ITAL_REC_HEADER + ldx + ldw + kmax doubles */
    int64_t* sel_ret;       /* [kmax + 1] as ital_select_fused; NULL: stop after the record (what ital_select_local leaves in
                               sel_record: several ranks exchange it and call ital_select_resolve) */
    double* sel_parts;      /* scratch: 3 doubles per block of the step's scoring launches (t = 1: n_cand / 256, t = 2:
                               n_cand / 32, t >= 3: n_cand / 256 + one per slab, rounded up) */
    int64_t sel_parts_len;  /* doubles in sel_parts */
    unsigned int* sel_counter; /* one zero-initialised word; left at zero by every call */
} ital_score_desc;

/* Scores every live candidate position: mi[p] = MI(batch + candidate p).
 * Replaces the Pool.map over AppendedMutualInformation.__call__, reference ital/ital.py:124-128, :504-529,
 * with MutualInformation._call_iter_all (:183-224), prob_rel (:345-383, scipy mvndst / norm.cdf) and
 * updated_prob_rel (:432-450, gp.updated_prediction / extend_inv). */
int ital_score_step(const ital_score_desc* d, hipStream_t stream);

/* Doubles of workspace that let ital_score_step(t >= 3) handle n_cand candidates in one slab (0 for t < 3). */
int64_t ital_score_workspace(int t, int64_t n_cand);

/* Sizes of the buffers a host owns, so that it need not re-derive them from the comments above (all in doubles):
 *   ital_record_len       a selection record = ITAL_REC_HEADER + ldx + ldw + kmax (sel_record, `record` of ital_select_*,
 *                         a row of records_all, rec_len of ital_select_exchange / ital_select_resolve);
 *   ital_round_workspace  `step.work` of a round of k steps over n_cand candidates when the host grants at most cap_doubles
 *                         (<= 0: no cap): one slab if that fits, else the cap (never below one candidate's need; 0 for k < 3);
 *   ital_sel_parts_len    `step.sel_parts` for the steps 1 .. k of such a round with a workspace of work_doubles: three per
 *                         block of the largest scoring launch plus one block per slab.
 * Pure host arithmetic: no device call, valid without a GPU. */
int ital_record_len(int ldx, int ldw, int kmax);
int64_t ital_round_workspace(int k, int64_t n_cand, int64_t cap_doubles);
int64_t ital_sel_parts_len(int k, int64_t n_cand, int64_t work_doubles);

/* One whole round of the perfect-user path -- fetch_unlabelled(k), reference ital/ital.py:98-134 -- enqueued by ONE call:
 * candidate-list upkeep, then for t = 1 .. k ital_score_step (ending with the selection inside its last launch: `step.sel_*`
 * must be set) and, for t < k, the new member's cross-covariance column (ital_cross_cov_cols out of the batch state into
 * C[t - 1]).  Nothing synchronises; the picks are ret[0 .. k), the status word ret[kmax].  At most ITAL_ROUND_MAX_CAND candidates
 * per rank. */
typedef struct ital_round_desc {
    int k;                          /* greedy steps */
    ital_score_desc step;           /* the steps' descriptor; t, mi (with mi_keep), seed, jump, jumppat, vk, ev_* are filled in per step */
    int seeds[ITAL_MAX_T + 1][6];   /* [t]: generator state before step t (t >= 3); the host advances its state by
                                       n_alive(t) * (2 << t) * ital_mvn_draws_per_call(t) per step, n_alive(t) = n_cand - (t - 1) */
    const long long* jump[ITAL_MAX_T + 1];      /* [t]: tables of ital_mvn_tables(t, ...) in device memory, t = 3 .. k */
    const long long* jumppat[ITAL_MAX_T + 1];
    const double* vk[ITAL_MAX_T + 1];
    void* ev_start[ITAL_MAX_T + 1]; /* optional hipEvent_t pairs around the lattice sums of step t */
    void* ev_stop[ITAL_MAX_T + 1];
    int64_t n_rows;                 /* rows of X / V / C (the covariance columns span all of them) */
    double var, length_scale;
    double* mi_keep;                /* optional [k][n_cand]: the scores of every step are kept (written there instead of step.mi) */
    /* candidate list upkeep before the first step: 0 none (cand / alive / ret[kmax] are ready); 1: alive[0 .. n_cand) = 1,
     * ret[kmax] = 0; 2: additionally step.cand = the entries of cand_prev[0 .. n_prev) whose alive flag is still set (the
     * previous round's list without its picks: reference retrieval_base.py:78-87 after an update with exactly that batch);
     * the survivors must number step.n_cand, else status |= 8 */
    int begin;
    const int32_t* cand_prev;
    int64_t n_prev;
    /* Several ranks (world > 1, or one rank driven through the exchange: world == 1 with a transport set): step.cand is this
     * rank's share of the list (list positions step.pos_offset + local position), every step ends with the rank's record
     * (the scoring launch's last workgroup), the exchange and ital_select_resolve; the candidate-list upkeep works on the
     * rank's own share (the survivors are the share without the picks that lay in this rank's rows).  The transport is
     * ital_select_exchange on nccl_comm, or the host's own `exchange` (same contract: records_all[w][rec_len] <- the record
     * of rank w on every rank, enqueued on / ordered with `stream`; return 0).  world == 0: one rank, no exchange. */
    int world;
    double* records_all;            /* [world][ITAL_REC_HEADER + step.sel_ldx + step.sel_ldw + step.batch.kmax] */
    void* nccl_comm;
    int (*exchange)(void* ctx, const double* record, double* records_all, int rec_len, hipStream_t stream);
    void* exchange_ctx;
} ital_round_desc;
int ital_fetch_round(const ital_round_desc* r, hipStream_t stream);

/* Local arg-extreme over the live positions + selection record for the exchange between ranks.
 * mode 0: first maximum, NaN wins (np.argmax, reference ital/ital.py:130); mode 1: first minimum (np.argmin,
 * ital/mcmi.py:77).  record = [value, list position, data index, mu, s2, |x|^2, rank, local position, the rank's
 * status word (*status, or 0 when status == NULL), reserved, x[ldx], V column[ldw], cross-covariances with the
 * members[kmax]]; record[1] < 0 when the rank has no live candidate.  List position of local position q:
 * gpos[q], or pos_offset + q when gpos == NULL.  work: >= 3*1024 doubles. */
int ital_select_local(const double* mi, const int32_t* cand, const uint8_t* alive, int64_t n_cand, int64_t pos_offset,
                      const int64_t* gpos, int64_t row_offset, int rank, int mode, const double* mu, const double* s2,
                      const double* X, const double* xnorm, int ldx, const double* V, int64_t ldv, int m, int ldw,
                      const double* C, int64_t ldc, int nprev, int kmax, const int* status, double* work, double* record,
                      hipStream_t stream);

/* Picks the winner among `world` records (same rule, lowest list position on ties), appends it to the batch
 * state as member `slot`, clears its alive flag on the owning rank and stores its data index in ret[slot];
 * ret[kmax] |= the OR of the status words of all records, so that every rank reads the same verdict (ret: kmax + 1).
 * Replaces mutual_information.append + del candidates[max_ind], reference ital/ital.py:131-132, :561-586. */
int ital_select_resolve(const double* records, int world, int rec_len, int rank, int mode, int slot,
                        ital_batch batch, uint8_t* alive, int64_t* ret, hipStream_t stream);

/* The exchange between ital_select_local and ital_select_resolve, for hosts that drive the ranks themselves: ONE
 * ncclAllGather (RCCL) per greedy step -- records_all[w][rec_len] receives the record of rank w, in rank order, on every
 * rank.  nccl_comm: this rank's ncclComm_t.  RCCL is looked up in the running process (a PyTorch host already carries one;
 * ital_amd's own learners go through torch.distributed, i.e. the same RCCL call) and opened by name otherwise.
 * Replaces the Pool.map gather of the per-candidate scores, reference ital/ital.py:124-130. */
int ital_select_exchange(const double* record, double* records_all, int rec_len, void* nccl_comm, hipStream_t stream);

/* What the exchange would run on: resolves RCCL exactly as ital_select_exchange does and asks it for the communicator's
 * size and this rank's number in it (ncclCommCount, ncclCommUserRank) -- no collective, nothing enqueued.  A host checks
 * with it, before the first exchange, that the communicator it is about to hand over orders the ranks as its own
 * bookkeeping does (ital_amd.sharding.raw_comm agrees on the outcome across the ranks before any of them uses it).
 * how (optional, how_len bytes): where RCCL was found.  -38: no RCCL in the process; -22: null communicator. */
int ital_exchange_info(void* nccl_comm, int* world, int* rank, char* how, int how_len);

/* Health of the communicator the exchanges run on: ncclCommGetAsyncError of the RCCL already in the process.  0: no error
 * (*async_error = ncclSuccess, or ncclInProgress on a non-blocking communicator); -5: RCCL's progress thread has recorded an
 * error (a peer died, a link failed: *async_error = the ncclResult_t, message in ital_last_error()); -38: no RCCL / no such
 * entry point.  Nothing is enqueued.  Collectives issued below torch.distributed are invisible to its watchdog: a host polls
 * this while it waits for the picks of a round (and bounds that wait: a rank that dies mid-round leaves the others inside
 * ncclAllGather) -- the failure detection the reference gets from multiprocessing.Pool raising in the parent when a worker
 * dies, reference ital/ital.py:124-126. */
int ital_exchange_error(void* nccl_comm, int* async_error);

/* ital_select_local + ital_select_resolve for ONE rank in a single launch (small problems are launch-latency bound).
 * Same semantics; `record` is scratch of ITAL_REC_HEADER + ldx + ldw + kmax doubles. */
int ital_select_fused(const double* mi, const int32_t* cand, uint8_t* alive, int64_t n_cand, int64_t pos_offset,
                      const int64_t* gpos, int64_t row_offset, int rank, int mode, const double* mu, const double* s2,
                      const double* X, const double* xnorm, int ldx, const double* V, int64_t ldv, int m, int ldw,
                      const double* C, int64_t ldc, int nprev, int slot, ital_batch batch, const int* status,
                      double* record, int64_t* ret, hipStream_t stream);

/* ---- MCMI[min] (pairwise objective) ---------------------------------------------------------------------- */

/* The candidate block of a fetch gathered out of this rank's rows in one launch: for j < nc and global sample index
 * cand[j] (device array) -- Xc[j][:] = X[cand[j] - row0][:], Vc[r][j] = V[r][cand[j] - row0] (r < m), xnc / muc / s2c[j] =
 * xnorm / mu / s2 of that row; a sample outside [row0, row0 + n_rows) contributes zeros (several ranks: the blocks are
 * summed).  Replaces the index arithmetic around self.candidates, reference ital/mcmi.py:57-66, :101-115 (the reference
 * indexes K_all / rel_mean with the candidate list). */
int ital_gather_block(const int64_t* cand, int64_t nc, int64_t row0, int64_t n_rows, const double* X, const double* xnorm,
                      int ldx, const double* V, int64_t ldv, int m, const double* mu, const double* s2, double* Xc,
                      double* Vc, int64_t ldc, double* xnc, double* muc, double* s2c, hipStream_t stream);

/* Dense posterior covariance block out[i][j] = k(a_i, b_j) - Va[:,i].Vb[:,j] between na and nb points (FP64 MFMA).
 * Replaces K_all[np.ix_(pred, ...)] - k_test^T K_inv k_test, reference ital/gp.py:226, :334-336, for the candidate
 * block MCMI_min scores against (reference ital/mcmi.py:115). */
int ital_cov_block(const double* Xa, const double* an, int64_t na, const double* Xb, const double* bn, int64_t nb,
                   int ldx, const double* Va, int64_t ldva, const double* Vb, int64_t ldvb, int m, double var,
                   double length_scale, double* out, int64_t ldo, hipStream_t stream);

/* out[i] (+)= sum_j |k(a_i, b_j) - Va[:,i].Vb[:,j]| without materialising the block (same FP64 MFMA tiles as
 * ital_cov_block; partial sums per column split in `work`, at least na doubles, summed in a fixed order).  With
 * `accumulate` the sums are added to out[] (column blocks of other ranks arriving one at a time).
 * Replaces the N x N product |alpha_diff . K_all[[T, i], :]|.mean() of EMOC.emoc_scores, reference
 * ital/baseline_methods.py:361-376: the model output change of candidate i is |(+-1 - mu_i) / (s2_i + noise)| times
 * the mean absolute posterior covariance of i with all samples. */
int ital_cov_abs_rowsum(const double* Xa, const double* an, int64_t na, const double* Xb, const double* bn, int64_t nb,
                        int ldx, const double* Va, int64_t ldva, const double* Vb, int64_t ldvb, int m, double var,
                        double length_scale, double* work, int64_t work_len, int accumulate, double* out,
                        hipStream_t stream);

typedef struct ital_mcmi_desc {
    int t;                  /* batch dimension of this greedy step */
    int64_t n_i;            /* candidates scored by this rank: block positions pos_offset .. pos_offset + n_i - 1 */
    int64_t pos_offset;
    int64_t n_all;          /* size of the candidate block (the objective sums over all of it) */
    const uint8_t* alive;   /* [n_i] 0 once picked */
    const double* mu;       /* [n_all] predictive mean of the block */
    const double* s2;       /* [n_all] predictive variance, NOT clamped */
    const double* cov;      /* [n_i][ld_cov] posterior covariance of the own candidates with the block */
    int64_t ld_cov;
    const double* C;        /* [t-1][ldc] posterior covariance of member b with the block */
    int64_t ldc;
    ital_batch batch;       /* bgpos = block position of each member */
    double noise, eps;
    double* ce;             /* [n_i] out: min over label patterns of the summed conditional entropy */
    double* work;           /* t >= 5 (required there, -22 without): ital_mcmi_workspace(t, n_i) doubles of device memory, any
                               content (the call clears the counters it keeps in it): the step runs as a preparation kernel
                               (W per candidate) and one workgroup per (candidate, group of 2^(t-3) label patterns).
                               t <= 4: unused (one kernel, workgroup per candidate) */
    int64_t work_doubles;
} ital_mcmi_desc;

/* ce[i] = min_r sum_j [q log(q+eps) + (1-q) log(1-q+eps)], q = P(candidate j irrelevant | batch + i labelled r).
 * Replaces the Pool.map over AppendedConditionalEntropy.__call__, reference ital/mcmi.py:69-75, :101-124
 * (updated_prediction(..., cov_mode='diag') over all candidates, gp.py:295-344, and scipy.stats.norm.cdf). */
int ital_mcmi_score_step(const ital_mcmi_desc* d, hipStream_t stream);
/* Doubles of workspace for the split form of ital_mcmi_score_step (0 for t < 5). */
int64_t ital_mcmi_workspace(int t, int64_t n_i);

/* One whole MCMI_min.fetch_unlabelled(k) round on ONE rank after the block is gathered -- reference ital/mcmi.py:66-79 --
 * enqueued by one call: [alive flags re-armed, ret[kmax] cleared,] the covariance of the candidates with the block
 * (ital_cov_block into step.cov), then for t = 1 .. k ital_mcmi_score_step, ital_select_fused (arg-min; `step.alive` is
 * written) and, for t < k, the picked member's covariance column (ital_cross_cov_cols into step.C[t - 1]).  Nothing
 * synchronises; the picks (block positions) are ret[0 .. k), the status word ret[kmax].  Several ranks drive the steps
 * themselves (record exchange between ital_select_local and ital_select_resolve). */
typedef struct ital_mcmi_round_desc {
    int k;
    ital_mcmi_desc step;        /* t is filled in per step; pos_offset must be 0 and n_i == n_all (one rank) */
    const double* Xc;           /* [n_all][ldx] feature rows of the block, xnc their squared norms (ital_gather_block) */
    const double* xnc;
    int ldx;
    const double* Vc;           /* [m][ldv] whitened columns of the block */
    int64_t ldv;
    int m, ldw;                 /* labelled samples; row stride of the batch state's VB */
    double var, length_scale;
    const int32_t* pos;         /* [n_all] 0, 1, 2, ...: block position of each candidate */
    const int* status;
    double* record;             /* scratch: ITAL_REC_HEADER + ldx + ldw + kmax doubles */
    int64_t* ret;               /* [kmax + 1] */
    int begin;                  /* 1: alive[0 .. n_i) = 1 and ret[kmax] = 0 first */
} ital_mcmi_round_desc;
int ital_mcmi_round(const ital_mcmi_round_desc* r, hipStream_t stream);

/* ---- general scorer: noisy user models, change-estimation subset ------------------------------------------- */
typedef struct ital_gscore_desc {
    int64_t n_cand;         /* candidate-list positions held by this rank */
    const int32_t* cand;    /* [n_cand] local row of each position */
    const uint8_t* alive;   /* [n_cand] */
    const double* mu;       /* [rows] */
    const double* s2;       /* [rows] NOT clamped */
    const double* C;        /* [nE][ldc] posterior covariance of base member e with every row */
    int64_t ldc;
    int64_t row_offset, pos_offset;
    const int64_t* gpos;    /* [n_cand] list position of every local position, or NULL: pos_offset + local position */
    /* base set E: the change-estimation subset followed by the picks outside it (subset_mode 1), or the picks */
    int nE;
    const int64_t* E_idx;   /* [nE] data indices */
    const int32_t* E_sort;  /* [nE] positions of E ordered by data index */
    const double* E_mu;     /* [nE] */
    const double* E_sig;    /* [nE][ldE] posterior covariance among E */
    int ldE;
    int n_picks;            /* picks so far (enumerated together with the candidate) */
    const int32_t* pick_pos;/* [n_picks] their positions in E, selection order */
    int subset_mode;        /* 0: MutualInformation._call_iter_all, 1: _call_iter_sub (reference ital/ital.py:183-275) */
    int fb_mode;            /* 0 perfect user, 1 label_prob >= 1 with mistakes, 2 general (reference ital/ital.py:300-342);
                               3 no simulated feedback: mi[] receives the batch entropy -sum_r p(r) log p(r) of
                               EntropySampling (reference ital/baseline_methods.py:243-287) */
    double label_prob, mistake_prob;
    int label_mode;         /* 0 mean, 1 optimistic, 2 pessimistic (subset_mode 0 only) */
    double noise, eps;
    double clip_cov;        /* 0 < clip_cov < 1: orthant probabilities above 5 dimensions factorise over the connected
                               components of |correlation| > clip_cov (reference ital/ital.py:360-362, :386-429, :590-616) */
    /* replay of mvndst's MVNUNI stream */
    int seed[6];            /* generator state before the first call of this greedy step */
    const long long* jump1; /* [ITAL_JUMP_BITS][18] transition matrices for 2^b uniforms */
    const double* vk;       /* [ITAL_GENERIC_MAX_DIM + 1][ITAL_GENERIC_MAX_DIM] Korobov generators of dimension n */
    int64_t draws_out;      /* uniforms one candidate outside E consumes */
    int64_t draws_in;       /* ... one live candidate that is a member of E */
    int n_in;
    const int64_t* in_pos;  /* [n_in] list positions of the live candidates inside E */
    int n_dead;
    const int64_t* dead_pos;/* [n_dead] list positions already picked */
    /* Monte-Carlo switches (reference ital/ital.py:293-297, :318-337): explicit sign patterns / feedback configurations
     * per candidate instead of the full enumeration; sampled by the host (numpy's global RNG, as the reference) */
    int mc_rel;             /* 0: enumerate the 2^(n_picks+1) patterns; > 0: patterns per candidate in rel_samples */
    const uint32_t* rel_samples; /* [n_cand][mc_rel] bit (n_picks - v) = sign of enumerated variable v */
    int mc_fb;              /* 0: enumerate; > 0: feedback configurations per pattern in fb_samples */
    const uint32_t* fb_samples;  /* [n_cand][patterns][mc_fb] low 16 bits: non-zero feedback (bit v), high 16: positive */
    const int64_t* draw_off;/* [n_cand] uniforms consumed before each candidate (replaces draws_out / draws_in when the
                               count per candidate varies: all-zero feedback samples are skipped without a call); or NULL */
    int64_t* draw_count;    /* non-NULL: counting pass -- only writes the uniforms every live candidate consumes
                               ([n_cand]; with clip_cov the count depends on the data) and scores nothing */
    double* mi;             /* [n_cand] out */
    int* status;
    double* work;           /* workspace in device memory (work_doubles doubles; ital_amd reuses the lattice scorer's).  Without a
                               subset and clip_cov, for 3 .. 16 variables, and with room for at least one candidate
                               (ital_score_generic_workspace: what one slab of all candidates takes; less means more slabs) the
                               step runs as a pipeline of kernels -- verdicts per call, preparation of the undecided calls,
                               lattice sums, combine -- the preparation of a chunk of calls under the lattice sums of the one
                               before (two internal streams, joined with `stream` on both sides); otherwise as one kernel that
                               does everything per candidate.  With 3 .. 6 variables the call WAITS ONCE PER SLAB on the host
                               (since round 6): the number of undecided calls is read back (4 bytes behind the verdict kernel)
                               so that only the chunks that hold entries are launched; everything else stays asynchronous */
    int64_t work_doubles;
    unsigned long long* pair_count; /* non-NULL: += the (Phi, Phi^-1) pairs of the lattice sums that were evaluated
                               (16 P(n-1) points x (n-1) pairs per evaluated call; instrumentation for the roofline) */
    int defer_join;         /* != 0 (pipeline, 7 .. 16 variables; ignored elsewhere): do NOT make `stream` wait for the internal
                               streams at the end of this call.  For a step scored in several calls over ranges of its
                               candidates (the host samples the patterns of range r + 1 while the GPU integrates range r):
                               the preparation of the next call's first slab then runs under the lattice sums of this one
                               instead of after them.  The results (mi) of a deferred call are complete only after a later
                               call of the same step WITHOUT the flag, or ital_score_generic_join, has returned; until then
                               the caller must not free, reuse or read what the call was given (mi, work, the samples) */
} ital_gscore_desc;

/* mi[p] = MI(batch + candidate p) for any user model / with a change-estimation subset.  Replaces
 * MutualInformation._call_iter_all / _call_iter_sub (reference ital/ital.py:183-275) with rel_iter (:278-291, full
 * enumeration), fb_iter (:300-342), likelihood (:453-481), prob_rel (:345-383) and updated_prob_rel (:432-450). */
int ital_score_generic(const ital_gscore_desc* d, hipStream_t stream);
/* Doubles of d->work with which the step of `d` (n_cand, nE, n_picks, fb_mode, mc_rel, mc_fb, subset_mode, clip_cov are read)
 * runs through the pipeline with all candidates in one slab; 0 when the step is not the pipeline's (single kernel, no
 * workspace needed). */
int64_t ital_score_generic_workspace(const ital_gscore_desc* d);
/* Makes `stream` wait for everything earlier ital_score_generic calls of this device left on the library's internal
 * streams (calls with defer_join; an error path that abandons a step half way).  Cheap when nothing is pending. */
int ital_score_generic_join(hipStream_t stream);

/* ---- context-style convenience layer ------------------------------------------------------------
 * A learner whose device buffers the LIBRARY owns (hipMalloc), for hosts that do not want to manage the buffers of the
 * descriptor entry points above themselves: the perfect-user path of ITAL -- fit, update, fetch_unlabelled(k <= ITAL_MAX_T)
 * with full sign-pattern enumeration, predict_stored.  Host code over the descriptor API of this same library (csrc/ctx.hip);
 * one context per learner, not thread-safe per context; the calls that return results to host memory synchronise `stream`.
 * Several ranks: rows sharded contiguously (rank r holds rows [n_total r / world, n_total (r + 1) / world)), one
 * ncclAllGather of a selection record per greedy step on `nccl_comm` (ital_select_local -> _exchange -> _resolve); with
 * world == 1 a non-NULL communicator sends the single rank through that same path.  ital_amd's Python learners do not use
 * this layer (their buffers are torch tensors). */
typedef struct ital_ctx ital_ctx;
/* capacity: most labelled samples (<= 0: 256).  Replaces ITAL.__init__ / ActiveRetrievalBase.__init__ + GaussianProcess.__init__,
 * reference ital/ital.py:15-81, ital/retrieval_base.py:7-31, ital/gp.py:99-139. */
int ital_ctx_create(int64_t n_total, int d, double length_scale, double var, double noise, int capacity, int rank, int world,
                    void* nccl_comm, ital_ctx** out);
int ital_ctx_destroy(ital_ctx* ctx);
/* rows: this rank's rows, [n_local][d] row-major (host memory, or device memory with on_device != 0); forgets every label.
 * Replaces ActiveRetrievalBase.fit / reset, reference ital/retrieval_base.py:34-61 (the dense kernel matrix of gp.py:128 is
 * never formed). */
int ital_ctx_fit(ital_ctx* ctx, const double* rows, int on_device, hipStream_t stream);
/* Labels c samples (global indices, y = +1 / -1).  Replaces ActiveRetrievalBase.update / GaussianProcess.update, reference
 * ital/retrieval_base.py:105-126, ital/gp.py:164-200.  -22 for a sample labelled before ("Cannot change feedback once
 * given."), -12 beyond the capacity, -38 on several ranks for samples that are neither local nor in the batch just fetched. */
int ital_ctx_update(ital_ctx* ctx, const int64_t* idx, const double* y, int c, hipStream_t stream);
/* picks[0 .. k) <- the batch (global indices, selection order); returns the number of picks (k clamped to the unlabelled
 * samples) or a negative code: -71 when the round needs the general scorer (ital_score_generic: duplicates in the batch,
 * noise too large for the limit verdicts).  Replaces ITAL.fetch_unlabelled, reference ital/ital.py:84-134. */
int ital_ctx_fetch(ital_ctx* ctx, int k, int64_t* picks, hipStream_t stream);
/* mean / variance [n_local] of this rank's rows (host memory; either may be NULL), the variance clamped at 0.  Replaces
 * GaussianProcess.predict_stored(cov_mode='diag') / rel_mean, reference ital/gp.py:203-232, ital/retrieval_base.py:58. */
int ital_ctx_predict_stored(ital_ctx* ctx, double* mean, double* variance, hipStream_t stream);
/* Rows this rank holds; *row0 (optional) <- the global index of its first row. */
int64_t ital_ctx_local_rows(const ital_ctx* ctx, int64_t* row0);

#ifdef __cplusplus
}
#endif
#endif /* ITAL_HIP_H */
