// General scorer, plain mode (no change-estimation subset, no clip_cov) with 1 .. 16 variables, as a pipeline of kernels
// through a workspace in HBM -- the noisy user models (reference ital/ital.py:300-342 `fb_iter`, :453-481 `likelihood`), the
// entropy baseline, the Monte-Carlo pattern switch up to batches of 16 (ital.py:293-297).  score_generic.hip's single
// kernel prepares, integrates and accumulates inside one wave per candidate; here every phase has the parallelism and the
// register allocation that suit it:
//
//   1, 2 variables (the first two greedy steps): closed forms -- gen_closed_kernel<T> (thread per (candidate, call)) + combine.
//
//   3 .. 6 variables (every shipped noisy-user configuration), "fast" form
//     gen_seed_kernel        thread per candidate          generator state at the candidate's first call
//     gen_verdict_kernel<T>  THREAD per (candidate, call)  decode, closed-form simulated update of the means and variances in
//                            (workgroup per candidate)     REGISTERS (compile-time T, the fed-back block as a mask), verdict
//                                                          from the standardised limits; undecided calls -> list U (one
//                                                          append per candidate)
//     gen_build_kernel<T>    THREAD per entry of U         the full updated covariance (registers), COVSRT (per-thread LDS
//                                                          slab), saturation test, the call's 8 lattices -> packed record,
//                                                          entry in the chunk's list (regular / linearly dependent variables)
//     gen_main_kernel<T|0>   WAVE per record               the lattice sum (FP64-VALU bound: the perfect-user evaluator)
//     gen_exact_kernel       wave per flagged record       label_estimation 'optimistic' / 'pessimistic' only: sums near 0 / 1
//                                                          again in MVKBRV's serial order (qmc_exact.h), every dimension
//     gen_combine_kernel     wave per candidate            the terms in the reference's order -> mi
//   With the general user at t = 4 only 240 of a candidate's 1296 calls need a lattice sum: verdicts for all calls with full
//   waves and no LDS, the expensive preparation (COVSRT, lattices) for the undecided ones only -- again with full waves (the
//   round-2/3 preparation did both per lane inside one wave per 64 calls: 18 % of the lanes busy in the expensive part, all
//   of it on runtime-indexed matrices in LDS; it took a third of the step's kernel time).  List U is worked off in chunks of
//   records (2^20 at most): gen_build of chunk c + 1 runs on a second stream under the lattice sums of chunk c.
//
//   7 .. 16 variables (sampled patterns, batches up to 16), "wide" form: gen_prep_kernel -- wave per candidate, lane per
//   call, matrices in LDS (the t x t algebra no longer fits registers) -- then the same lattice-sum and combine kernels.
//   Calls per candidate are few there (2 t c) and the lattice sums of 6 .. 15 dimensions dwarf the preparation.
//
// Record of a call to integrate (doubles): meta word, id (candidate of the slab * calls + call), packed factor with
// diagonal (n (n+1) / 2) and limits (n) as the evaluators read them, then the call's 8 shifted lattices packed: 8 (n-1) shifts
// as MVNUNI's 32-bit integers, the 8 permuted generator vectors as 4-bit indices (a 64-bit word per shift).
// the kernels of this unit call each of the larger helpers once: inlined (an out-of-line callee saves registers on a stack)
#define ITAL_GEN_FN __device__ __forceinline__
#include <algorithm>

#include "gen_common.h"
#include "qmc_exact.h"

namespace ital {

constexpr int GEN_FLAG_EXACT = 128;   // record flag set by the lattice sum: recompute (qmc_exact.h)

struct GPipe {
    int64_t slab_lo, slab_n;   // candidate positions [slab_lo, slab_lo + slab_n) of this slab
    int total;                 // calls per candidate
    int npat, cpp;             // patterns, calls per pattern (1 prior + feedback configurations)
    int n;                     // variables of every call (n_picks + 1)
    int R;                     // doubles per record
    int lat;                   // offset of the packed lattices inside a record
    double* meta;              // [slab_n][total][2]: (flags | n << 8 | infi << 16 | closes << 40, value)
    int* cstate;               // [slab_n][6] generator state at the candidate's first call (fast form)
    unsigned int* listU;       // [slab_n * total] undecided calls (fast form)
    unsigned int* countU;      // their number
    // chunk of records being built / integrated
    double* recs;              // [cap][R]
    unsigned int* list;        // [cap] record indices: regular calls from the front, linearly dependent ones from the back
    unsigned int* count;       // [2] entries from the front / from the back
    unsigned int cap;          // records in this buffer
    unsigned int chunk_lo;     // fast form: the chunk covers entries [chunk_lo, chunk_lo + cap) of list U
    int nsplit;                // wide form: waves a candidate's calls are spread over in the preparation
    unsigned char* redo;       // wide form, perfect user: [slab_n] candidates gen_prep_pu_kernel left to gen_prep_kernel (NULL: all)
    int npre;                  // prior calls per pattern: 1, or 2 with a change-estimation subset (the enumerated variables alone,
                               // then all of U: reference ital.py:227-275)
    int mixed;                 // the calls of a step differ in their number of variables (subset mode): gen_main_kernel<T> takes
                               // the records of T variables out of the list, one launch per dimension that occurs
};

__device__ __forceinline__ long long pack_meta(int flags, int n, unsigned infi, unsigned closes) {
    return (long long)(flags & 0xff) | ((long long)(n & 0xff) << 8) | ((long long)(infi & 0xffffffu) << 16) |
           ((long long)(closes & 0xffffffu) << 40);
}

__device__ __forceinline__ int calls_feedbacks(const ital_gscore_desc& d, int nr) {
    return d.fb_mode == 3 ? 0 : (d.fb_mode == 0 ? 1 : (d.mc_fb > 0 ? d.mc_fb : (d.fb_mode == 1 ? (1 << nr) : pow3(nr) - 1)));
}

// Generator state `base` advanced by `before` uniforms (one 3x3 product mod m per set bit).
__device__ __forceinline__ MrgState mrg_jump(const ital_gscore_desc& d, MrgState st, uint64_t before) {
    for (int bit = 0; before != 0 && bit < ITAL_JUMP_BITS; bit++, before >>= 1)      // (the table has ITAL_JUMP_BITS rows: the
        if (before & 1u) mrg_apply(st, d.jump1 + bit * 18);                          // host rejects larger offsets, -22)
    return st;
}

// The 8 lattices of a call of dimension n from generator state `sti`, packed into `out` (4 (n-1) + 8 doubles): the shifts as
// MVNUNI's 32-bit integers, the generator vector after DKSMRC's accumulated random transpositions as 4-bit indices into
// vk[n][.], one 64-bit word per shift.
// gen: n - 1 doubles of scratch (LDS: indexed at run time).
static __device__ ITAL_GEN_NOINLINE void make_lattice_packed(MrgState sti, int n, double* gen, double* __restrict__ out) {
    MrgStateF st = mrg_to_f(sti);
    const int ndim = n - 1;
    for (int j = 0; j < ndim; j++) gen[j] = (double)j;
    unsigned int* shifts = reinterpret_cast<unsigned int*>(out);
    unsigned long long* perm = reinterpret_cast<unsigned long long*>(out + 4 * ndim);
    for (int sft = 0; sft < 8; sft++) {
        for (int j = 1; j <= ndim - 1; j++) {
            const double u = mrg_next_f(st);
            const int jp = (int)(j + u * (ndim + 1 - j));
            const double xt = gen[j - 1];
            gen[j - 1] = gen[jp - 1];
            gen[jp - 1] = xt;
        }
        unsigned long long w = 0;                      // ndim <= 15 indices below 16: four bits each
        for (int j = 0; j < ndim; j++) w |= (unsigned long long)(unsigned int)gen[j] << (4 * j);
        perm[sft] = w;
        for (int j = 0; j < ndim; j++) shifts[sft * ndim + j] = (unsigned int)mrg_next_z(st);
    }
}

// Slab (packed factor with diagonal, limits) of a call that goes to the compile-time evaluators into its record: those take
// every variable as bounded above (ITAL_QMC_FLIP, qmc_common.h) -- a variable bounded below enters negated: its limit, its
// row and its column of the factor change sign (the lattice sum moves its shifts by 1/2 when it unpacks them).
__device__ __forceinline__ void write_slab(int n, const double* slab, unsigned fl, double* __restrict__ rec) {
    int q = 0;
    for (int v = 0; v < n; v++)
        for (int j = 0; j <= v; j++, q++) rec[q] = (((fl >> v) ^ (fl >> j)) & 1u) ? -slab[q] : slab[q];
    for (int v = 0; v < n; v++) rec[q + v] = ((fl >> v) & 1u) ? -slab[q + v] : slab[q + v];
}

// ------------------------------------------------------------------------------------------------ fast form, 3 .. 6 variables
// Joint prior of U = (batch so far, candidate) of the candidate at local row `row`: means and packed lower covariance.
template <int T>
__device__ __forceinline__ void load_prior(const ital_gscore_desc& d, int row, double (&mu)[T], double (&Sg)[T * (T + 1) / 2]) {
#pragma unroll
    for (int v = 0; v < T - 1; v++) {
        mu[v] = d.E_mu[v];
#pragma unroll
        for (int j = 0; j <= v; j++) Sg[v * (v + 1) / 2 + j] = d.E_sig[v * d.ldE + j];
        Sg[(T - 1) * T / 2 + v] = d.C[(int64_t)v * d.ldc + row];
    }
    mu[T - 1] = d.mu[row];
    Sg[(T - 1) * T / 2 + T - 1] = d.s2[row];                          // not clamped (gp.py:254)
}

// Closed-form posterior after the simulated update with the feedback set Fm (bit u = variable u of U fed back; Fp: with +1)
//     W = (Sigma_FF + s I)^-1, g = W (f - mu_F),  mu'_F = f - s g,  mu'_a = mu_a + Sigma_aF g,
//     Sigma'_FF = s (I - s W),  Sigma'_aF = s Sigma_aF W,  Sigma'_ab = Sigma_ab - Sigma_aF W Sigma_Fb
// (gen_common.h prepare_call, the same sums in the same order) for a compile-time number of variables, everything in
// registers: the fed-back block is not compacted -- the matrix that is factored carries unit rows / columns for the
// variables outside F, which leaves W on F exactly as the compacted factorisation gives it (the extra terms are exact
// zeros) and lets every loop run over 0 .. T-1.  FULL = false: means and variances only (what the verdict needs).
template <int T, bool FULL>
__device__ __forceinline__ void masked_update(const double (&mu)[T], const double (&Sg)[T * (T + 1) / 2], unsigned Fm, unsigned Fp,
                                              double s, double (&mean)[T], double (&cv)[T * (T + 1) / 2]) {
    auto S = [&](int a, int b) -> double { return a >= b ? Sg[a * (a + 1) / 2 + b] : Sg[b * (b + 1) / 2 + a]; };
    bool inF[T];
#pragma unroll
    for (int u = 0; u < T; u++) inF[u] = (Fm >> u) & 1u;
    double L[T][T];      // lower factor, then (in X) its inverse
#pragma unroll
    for (int i = 0; i < T; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double v = (inF[i] && inF[j]) ? S(i, j) + (i == j ? s : 0.0) : (i == j ? 1.0 : 0.0);
#pragma unroll
            for (int q = 0; q < j; q++) v -= L[i][q] * L[j][q];
            L[i][j] = (i == j) ? sqrt(v) : v / L[j][j];
        }
    double X[T][T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        X[j][j] = 1.0 / L[j][j];
#pragma unroll
        for (int i = j + 1; i < T; i++) {
            double sm = 0;
#pragma unroll
            for (int q = j; q < i; q++) sm += L[i][q] * X[q][j];
            X[i][j] = -sm / L[i][i];
        }
    }
    double W[T][T];      // W = X^T X on F, zero elsewhere
#pragma unroll
    for (int i = 0; i < T; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double w = 0;
#pragma unroll
            for (int k = i; k < T; k++) w += X[k][j] * X[k][i];
            w = (inF[i] && inF[j]) ? w : 0.0;
            W[i][j] = w;
            W[j][i] = w;
        }
    double gv[T];
#pragma unroll
    for (int a = 0; a < T; a++) {
        double acc = 0;
#pragma unroll
        for (int b = 0; b < T; b++) {
            const double fb = ((Fp >> b) & 1u) ? 1.0 : -1.0;
            acc += W[a][b] * (fb - mu[b]);
        }
        gv[a] = acc;
    }
#pragma unroll
    for (int a = 0; a < T; a++) {
        double m = mu[a];
#pragma unroll
        for (int f = 0; f < T; f++) m += S(a, f) * gv[f];              // gv is zero outside F
        mean[a] = inF[a] ? (((Fp >> a) & 1u) ? 1.0 : -1.0) - s * gv[a] : m;
    }
    // SW[f][b] = sum_g W[f][g] Sigma[g][b]
    double SW[T][T];
#pragma unroll
    for (int f = 0; f < T; f++)
#pragma unroll
        for (int b = 0; b < T; b++) {
            double inner = 0;
#pragma unroll
            for (int g2 = 0; g2 < T; g2++) inner += W[f][g2] * S(g2, b);
            SW[f][b] = inner;
        }
#pragma unroll
    for (int a = 0; a < T; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) {
            if (!FULL && a != b) continue;
            double both = s * ((a == b ? 1.0 : 0.0) - s * W[a][b]);
            double one_b = 0, one_a = 0, none = 0;     // b in F only / a in F only / neither
#pragma unroll
            for (int f = 0; f < T; f++) {
                one_b += S(a, f) * W[f][b];
                one_a += S(b, f) * W[f][a];
                none += S(a, f) * SW[f][b];
            }
            const double v = (inF[a] && inF[b]) ? both : (inF[b] ? s * one_b : (inF[a] ? s * one_a : S(a, b) - none));
            cv[a * (a + 1) / 2 + b] = v;
        }
}

// The sign every variable of U takes in a call and its feedback set, as bits over the positions of U.
template <int T>
__device__ __forceinline__ void call_masks(const CallInfo& ci, const int (&ipos)[T], unsigned& relU, unsigned& Fm, unsigned& Fp) {
    relU = 0; Fm = 0; Fp = 0;
#pragma unroll
    for (int v = 0; v < T; v++) {
        relU |= ((ci.pat >> (T - 1 - v)) & 1u) << ipos[v];
        if (ci.kind == K_UPDATED && ((ci.fnz >> v) & 1u)) {
            Fm |= 1u << ipos[v];
            if ((ci.fpos >> v) & 1u) Fp |= 1u << ipos[v];
        }
    }
}

struct GSeed {
    ital_gscore_desc d;
    int64_t slab_lo, slab_n;
    int* cstate;
};

// Generator state at the first call of every candidate of the slab: the step's seed advanced by what the candidates in
// front of it in the reference's serial order consume.
__global__ __launch_bounds__(256) void gen_seed_kernel(GSeed s) {
    const ital_gscore_desc& d = s.d;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.slab_n) return;
    const int64_t p = s.slab_lo + i;
    if (!d.alive[p]) return;
    const int64_t gpos = d.gpos ? d.gpos[p] : d.pos_offset + p;
    int64_t before = gpos;
    for (int q = 0; q < d.n_dead; q++) before -= (d.dead_pos[q] < gpos) ? 1 : 0;
    uint64_t off = (uint64_t)before * (uint64_t)d.draws_out;
    if (d.draw_off) off = (uint64_t)d.draw_off[p];
    MrgState rng = {d.seed[0], d.seed[1], d.seed[2], d.seed[3], d.seed[4], d.seed[5]};
    rng = mrg_jump(d, rng, off);
    int* sp = s.cstate + i * 6;
    sp[0] = rng.x10; sp[1] = rng.x11; sp[2] = rng.x12; sp[3] = rng.x20; sp[4] = rng.x21; sp[5] = rng.x22;
}

// Thread per (candidate, call): is the call's orthant probability decided by its standardised limits alone?  One workgroup
// per candidate walks its calls in groups of blockDim.x (the host picks the width that wastes the fewest lanes); the
// undecided ones are collected in LDS and appended to list U with ONE atomic per candidate (a wave-level append -- 190 000
// atomics on one address per noisy-user step -- took 1.9 ms of this kernel's 2.2).  With sampled feedback all-zero samples
// make no call and draw nothing: the stream offset of a call is the number of calls made before it, counted along the walk.
template <int T>
__global__ __launch_bounds__(256) void gen_verdict_kernel(ital_gscore_desc d, GPipe g) {
    constexpr int STAGE = 2048;                           // undecided ids collected before a flush (>= 8 groups)
    __shared__ unsigned int s_ids[STAGE];
    __shared__ unsigned int s_made[4], s_und[4];
    __shared__ unsigned int s_base;
    const int64_t i = blockIdx.x;
    const int64_t p = g.slab_lo + i;
    if (!d.alive[p]) return;
    const int row = d.cand[p];
    double mu[T], Sg[T * (T + 1) / 2];
    load_prior<T>(d, row, mu, Sg);
    int ipos[T];
#pragma unroll
    for (int v = 0; v < T; v++) ipos[v] = v < d.n_picks ? d.pick_pos[v] : d.nE;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    constexpr unsigned DRAWS = 8 * (2 * (T - 1) - 1);     // uniforms of every call that is made (integrated or not)
    double* meta = g.meta + (size_t)i * g.total * 2;
    unsigned int made_before = 0, staged = 0;             // calls made / undecided ids staged so far (the same in every thread)
    auto flush = [&]() {                                  // (all threads)
        if (threadIdx.x == 0) s_base = atomicAdd(g.countU, staged);
        __syncthreads();
        for (unsigned int j = threadIdx.x; j < staged; j += blockDim.x) g.listU[s_base + j] = s_ids[j];
        __syncthreads();
        staged = 0;
    };
    for (int c0 = 0; c0 < g.total; c0 += (int)blockDim.x) {
        const int call = c0 + (int)threadIdx.x;
        const bool mine = call < g.total;
        int flags = 16;
        if (mine) {
            const CallInfo ci = decode_call(d, p, call, g.cpp, 1, T, g.npat);
            if (ci.kind != K_SKIP) {
                unsigned relU, Fm, Fp;
                call_masks<T>(ci, ipos, relU, Fm, Fp);
                double mean[T], cv[T * (T + 1) / 2];
                masked_update<T, false>(mu, Sg, Fm, Fp, d.noise, mean, cv);
                // early_decision (gen_common.h) in the natural order of the variables: it does not depend on their order
                const double thr = 37.0 + 9.0 * sqrt((double)(T - 1));
                bool all_full = true, any_empty = false;
#pragma unroll
                for (int a = 0; a < T; a++) {
                    const double lim = -mean[a] / sqrt(cv[a * (a + 1) / 2 + a]);
                    const double l = ((relU >> a) & 1u) ? lim : -lim;
                    if (l > thr) any_empty = true;
                    if (!(l < -thr)) all_full = false;
                }
                flags = ITAL_GEN_EARLY ? (any_empty ? 4 : (all_full ? 2 : 0)) : 0;
            }
        }
        const bool made = mine && !(flags & 16);
        const bool undecided = mine && flags == 0;
        const unsigned long long mm = __ballot(made), um = __ballot(undecided);
        if (lane == 0) { s_made[wid] = (unsigned int)__popcll(mm); s_und[wid] = (unsigned int)__popcll(um); }
        __syncthreads();
        unsigned int rank_made = made_before + (unsigned int)__popcll(mm & ((1ull << lane) - 1ull));
        unsigned int slot = staged + (unsigned int)__popcll(um & ((1ull << lane) - 1ull));
        unsigned int made_all = 0, und_all = 0;
        for (int w = 0; w < nwave; w++) {
            if (w < wid) { rank_made += s_made[w]; slot += s_und[w]; }
            made_all += s_made[w];
            und_all += s_und[w];
        }
        if (mine) {
            meta[2 * call] = __longlong_as_double(pack_meta(flags, T, 0, 0));
            // decided: the value; undecided: where the call's uniforms start in the candidate's stretch of the stream
            meta[2 * call + 1] = undecided ? (double)((uint64_t)rank_made * DRAWS) : ((flags & 2) ? 1.0 : 0.0);
            if (undecided) s_ids[slot] = (unsigned int)(i * g.total + call);
        }
        made_before += made_all;
        staged += und_all;
        __syncthreads();
        if (staged + blockDim.x > (unsigned int)STAGE) flush();
    }
    if (staged) flush();
}

// One or two variables (the first two greedy steps): every call is a closed form -- norm.cdf / Genz's BVU, exactly as
// prepare_call / finish_call (gen_common.h) evaluate them -- so the step is this kernel and the combine: thread per
// (candidate, call), one workgroup per candidate walking its calls; no lattice, no stream, no list.
template <int T>
__global__ __launch_bounds__(256) void gen_closed_kernel(ital_gscore_desc d, GPipe g) {
    static_assert(T == 1 || T == 2, "closed forms exist for one and two variables");
    const int64_t i = blockIdx.x;
    const int64_t p = g.slab_lo + i;
    if (!d.alive[p]) return;
    const int row = d.cand[p];
    double mu[T], Sg[T * (T + 1) / 2];
    load_prior<T>(d, row, mu, Sg);
    int ipos[T];
#pragma unroll
    for (int v = 0; v < T; v++) ipos[v] = v < d.n_picks ? d.pick_pos[v] : d.nE;
    // two variables: an updated call takes them in the order of their data indices (ital.py:448)
    bool swap = false;
    if (T == 2) swap = d.row_offset + row < d.E_idx[0];
    double* meta = g.meta + (size_t)i * g.total * 2;
    for (int call = (int)threadIdx.x; call < g.total; call += (int)blockDim.x) {
        const CallInfo ci = decode_call(d, p, call, g.cpp, 1, T, g.npat);
        int flags = 16;
        double value = 0.0;
        if (ci.kind != K_SKIP) {
            unsigned relU, Fm, Fp;
            call_masks<T>(ci, ipos, relU, Fm, Fp);
            double mean[T], cv[T * (T + 1) / 2];
            masked_update<T, true>(mu, Sg, Fm, Fp, d.noise, mean, cv);
            flags = 1;
            if (T == 1) {
                double var = cv[0];
                if (ci.kind != K_UPDATED) var = fmax(0.0, var);          // first step: predict_stored 'diag' (gp.py:229, ital.py:558)
                const double p_irr = norm_cdf0(mean[0], sqrt(var));       // prob_rel, ital.py:364-369
                value = (relU & 1u) ? 1.0 - p_irr : p_irr;
            } else {
                const int a0 = (swap && ci.kind == K_UPDATED) ? 1 : 0, a1 = 1 - a0;       // variables at positions 0, 1 of the call
                const double sd0 = sqrt(cv[a0 * (a0 + 1) / 2 + a0]), sd1 = sqrt(cv[a1 * (a1 + 1) / 2 + a1]);
                const double l0 = -mean[a0] / sd0, l1 = -mean[a1] / sd1;
                const double rho = cv[T > 1 ? 1 : 0] / (sd1 * sd0);
                value = bvn_orthant(l0, l1, (relU >> a0) & 1u, (relU >> a1) & 1u, rho);
            }
        }
        meta[2 * call] = __longlong_as_double(pack_meta(flags, T, 0, 0));
        meta[2 * call + 1] = value;
    }
}

// Thread per undecided call: the standardised problem after the simulated update (registers), COVSRT in the thread's LDS
// slab, saturation test; a call that needs its lattice sum gets a record (its 8 lattices drawn at the stream position the
// serial reference reaches) and an entry in the chunk's list.
template <int T>
__global__ __launch_bounds__(256) void gen_build_kernel(ital_gscore_desc d, GPipe g) {
    extern __shared__ double lds_all[];
    constexpr int NCOV = T * (T + 1) / 2, STRIDE = (NCOV + 2 * T) | 1;
    const int lane = threadIdx.x & 63;
    const unsigned int nU = *g.countU;
    const unsigned int r = blockIdx.x * blockDim.x + threadIdx.x;     // record of this chunk
    const unsigned int e = g.chunk_lo + r;
    const bool mine = r < g.cap && e < nU;
    bool integrate = false, regular = false;
    unsigned infi = 0, closes = 0;      // limit types (bit a: variable at position a bounded below), rows that close a group
    int flags = 0;                      // 2 / 4: the integrand is identically 1 / 0
    double* slab = lds_all + (size_t)threadIdx.x * STRIDE;
    unsigned int id = 0;
    int64_t i = 0;
    int call = 0;
    if (mine) {
        id = g.listU[e];
        i = id / (unsigned int)g.total;
        call = (int)(id - (unsigned int)i * (unsigned int)g.total);
        const int64_t p = g.slab_lo + i;
        const int row = d.cand[p];
        double mu[T], Sg[NCOV];
        load_prior<T>(d, row, mu, Sg);
        int ipos[T];
#pragma unroll
        for (int v = 0; v < T; v++) ipos[v] = v < d.n_picks ? d.pick_pos[v] : d.nE;
        const CallInfo ci = decode_call(d, p, call, g.cpp, 1, T, g.npat);
        unsigned relU, Fm, Fp;
        call_masks<T>(ci, ipos, relU, Fm, Fp);
        double mean[T], cv[NCOV];
        masked_update<T, true>(mu, Sg, Fm, Fp, d.noise, mean, cv);
        // order of the variables inside the call: natural for a prior call (ital.py:373-383), by data index for an updated
        // one (updated_prob_rel sorts the ids, ital.py:448): position of variable u
        int at[T];
        if (ci.kind == K_UPDATED) {
            const int64_t gi = d.row_offset + row;
            int rank = 0;
            for (int q = 0; q < d.nE; q++) rank += (d.E_idx[q] < gi) ? 1 : 0;
            for (int sidx = 0; sidx < d.nE; sidx++) {
                const int u = d.E_sort[sidx];
#pragma unroll
                for (int v = 0; v < T - 1; v++)
                    if (u == v) at[v] = sidx < rank ? sidx : sidx + 1;
            }
            at[T - 1] = rank;
        } else {
#pragma unroll
            for (int v = 0; v < T; v++) at[v] = v;
        }
        double* cov = slab;
        double* lim = slab + NCOV;
        double* y = lim + T;
        double sd[T];
#pragma unroll
        for (int u = 0; u < T; u++) {
            sd[u] = sqrt(cv[u * (u + 1) / 2 + u]);
            lim[at[u]] = -mean[u] / sd[u];
            infi |= ((relU >> u) & 1u) << at[u];
            cov[pidx(at[u], at[u])] = 1.0;
        }
#pragma unroll
        for (int u = 1; u < T; u++)
#pragma unroll
            for (int v = 0; v < u; v++) {
                const int a = at[u] > at[v] ? at[u] : at[v], b = at[u] > at[v] ? at[v] : at[u];
                // (y[a] * y[b] of prepare_call: the standard deviations in the call's order, larger position first)
                cov[pidx(a, b)] = cv[u * (u + 1) / 2 + v] / (at[u] > at[v] ? sd[u] * sd[v] : sd[v] * sd[u]);
            }
        flags = ITAL_GEN_EARLY ? early_decision(T, lim, infi) : 0;
        if (!flags) {
            infi = covsrt_n(T, cov, lim, y, infi);
            closes = group_layout(T, cov);
            flags = saturation_n(T, cov, lim, infi);
        }
        integrate = !(flags & 6);
        regular = integrate && closes == (1u << T) - 1u;
    }
    // list slots: one atomic per workgroup and list end (regular calls from the front, the others from the back)
    __shared__ unsigned int s_reg[4], s_dep[4], s_lbase, s_cbase;
    const int wid = threadIdx.x >> 6;
    const unsigned long long em = __ballot(regular), cm = __ballot(integrate && !regular);
    if (lane == 0) { s_reg[wid] = (unsigned int)__popcll(em); s_dep[wid] = (unsigned int)__popcll(cm); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int nr_ = s_reg[0] + s_reg[1] + s_reg[2] + s_reg[3], nd_ = s_dep[0] + s_dep[1] + s_dep[2] + s_dep[3];
        s_lbase = nr_ ? atomicAdd(g.count, nr_) : 0u;
        s_cbase = nd_ ? atomicAdd(g.count + 1, nd_) : 0u;
    }
    __syncthreads();
    unsigned int lbase = s_lbase, cbase = s_cbase;
    for (int w = 0; w < wid; w++) { lbase += s_reg[w]; cbase += s_dep[w]; }
    if (!mine) return;
    double* meta = g.meta + ((size_t)i * g.total + call) * 2;
    if (integrate) {
        double* rec = g.recs + (size_t)r * g.R;
        const uint64_t before = (uint64_t)meta[1];
        rec[0] = __longlong_as_double(pack_meta(flags, T, infi, closes));
        rec[1] = (double)id;
        write_slab(T, slab, (regular && ITAL_QMC_FLIP) ? infi : 0u, rec + 2);
        const int* sp = g.cstate + i * 6;
        MrgState st = {sp[0], sp[1], sp[2], sp[3], sp[4], sp[5]};
        make_lattice_packed(mrg_jump(d, st, before), T, slab, rec + g.lat);        // the slab is free now
        if (regular) g.list[lbase + (unsigned int)__popcll(em & ((1ull << lane) - 1ull))] = r;
        else g.list[g.cap - 1u - cbase - (unsigned int)__popcll(cm & ((1ull << lane) - 1ull))] = r;
        meta[0] = __longlong_as_double(pack_meta(flags, T, infi, closes));
    } else {
        meta[0] = __longlong_as_double(pack_meta(flags, T, infi, closes));
        meta[1] = (flags & 2) ? 1.0 : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------ wide form, 7 .. 16 variables
// Wave per candidate (ITAL_GEN_PREP_SPLIT waves when its calls fill several passes), lane per call: decode, closed-form
// update, verdict, COVSRT, lattices -- the matrices in the lane's LDS slab (gen_common.h prepare_call).
__global__ __launch_bounds__(128) void gen_prep_kernel(GArgs a, GPipe g) {
    extern __shared__ double lds_all[];
    const ital_gscore_desc& d = a.d;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t item = (int64_t)blockIdx.x * 2 + wid;
    const int64_t i = item / g.nsplit;
    const int part = (int)(item - i * g.nsplit);
    if (i >= g.slab_n) return;
    const int64_t p = g.slab_lo + i;
    if (!d.alive[p]) return;
    if (g.redo && !g.redo[i]) return;        // prepared by gen_prep_pu_kernel already
    double* W = lds_all + (size_t)wid * a.wave_doubles;
    double* muU = W;
    const int ldS = a.ldS;
    double* SigU = muU + ldS;
    int* usort = reinterpret_cast<int*>(SigU + ldS * ldS);
    int* ipos = usort + GN;
    double* slabs = SigU + ldS * ldS + (GN + GR + 1) / 2;

    const int row = d.cand[p];
    const int64_t gi = d.row_offset + row;
    const int nE = d.nE;
    const bool subset = d.subset_mode != 0;
    // plain mode: U = batch so far + candidate.  With a change-estimation subset U = E (subset + picks), plus the candidate
    // unless it is a member of E itself (score_generic_kernel, the single-kernel form, is the statement this follows)
    int epos = -1;
    if (subset)
        for (int e = 0; e < nE; e++)
            if (d.E_idx[e] == gi) epos = e;
    const int nU = epos >= 0 ? nE : nE + 1;
    const int cpos = epos >= 0 ? epos : nE;
    const int nr = d.n_picks + 1;
    for (int idx = lane; idx < nU * nU; idx += 64) {
        const int r = idx / nU, c = idx - r * nU;
        double v;
        if (r < nE && c < nE) v = d.E_sig[r * d.ldE + c];
        else if (r == c) v = d.s2[row];                         // not clamped (gp.py:254)
        else v = d.C[(int64_t)(r < c ? r : c) * d.ldc + row];
        SigU[r * ldS + c] = v;
    }
    for (int e = lane; e < nU; e += 64) muU[e] = e < nE ? d.E_mu[e] : d.mu[row];
    if (lane == 0) {
        if (epos >= 0) {
            for (int sidx = 0; sidx < nE; sidx++) usort[sidx] = d.E_sort[sidx];
        } else {
            int rank = 0;
            for (int e = 0; e < nE; e++) rank += (d.E_idx[e] < gi) ? 1 : 0;
            for (int sidx = 0; sidx < nU; sidx++)
                usort[sidx] = sidx < rank ? d.E_sort[sidx] : (sidx == rank ? nE : d.E_sort[sidx - 1]);
        }
        for (int v = 0; v < nr; v++) ipos[v] = v < d.n_picks ? d.pick_pos[v] : cpos;
    }
    const int total = g.total;
    // this wave's share of the candidate's calls: whole passes of a.chunk calls
    const int npass = (total + a.chunk - 1) / a.chunk;
    const int pass_lo = (int)((int64_t)npass * part / g.nsplit), pass_hi = (int)((int64_t)npass * (part + 1) / g.nsplit);
    // stream position of this candidate in the reference's serial order, then of the share's first call (with more than one
    // share every call draws the same 8 (2 (nU - 1) - 1) uniforms: no skipped samples)
    MrgState rng = {d.seed[0], d.seed[1], d.seed[2], d.seed[3], d.seed[4], d.seed[5]};
    {
        const int64_t gpos = d.gpos ? d.gpos[p] : d.pos_offset + p;
        int64_t before = gpos, n_in = 0;
        for (int q = 0; q < d.n_dead; q++) before -= (d.dead_pos[q] < gpos) ? 1 : 0;
        for (int q = 0; q < d.n_in; q++) n_in += (d.in_pos[q] < gpos) ? 1 : 0;      // members of E ahead consume draws_in each
        uint64_t off = (uint64_t)(before - n_in) * (uint64_t)d.draws_out + (uint64_t)n_in * (uint64_t)d.draws_in;
        if (d.draw_off) off = (uint64_t)d.draw_off[p];
        // (several shares per candidate only where every call draws the same: never with a subset, nsplit == 1 there)
        off += (uint64_t)pass_lo * (uint64_t)a.chunk * (uint64_t)(nU >= 3 ? 8 * (2 * (nU - 1) - 1) : 0);
        rng = mrg_jump(d, rng, off);
    }
    wave_sync();
    const bool clamp_prior = !subset && nr == 1;
    double* meta = g.meta + (size_t)i * total * 2;
    for (int chunk0 = pass_lo * a.chunk; chunk0 < pass_hi * a.chunk && chunk0 < total; chunk0 += a.chunk) {
        Prep pp;
        pp.n = 0; pp.infi = 0; pp.flags = 16; pp.value = 0; pp.closes = 0; pp.ng = 0; pp.gdraws = 0;
        const int call = chunk0 + lane;
        const bool mine = lane < a.chunk && call < total;
        double* slab = slabs + (size_t)lane * a.stride;
        if (mine) {
            const CallInfo ci = decode_call(d, p, call, g.cpp, g.npre, nr, g.npat);
            if (ci.kind != K_SKIP)
                pp = prepare_call<false>(prep_opts(d), ci, nU, nr, ldS, muU, SigU, usort, ipos, clamp_prior, slab, slab + a.slab, nullptr);
        }
        // every dimension >= 3 call (evaluated or saturated) takes 8*(2*NDIM-1) uniforms from MVNUNI: lane l jumps ahead by
        // what the calls before it in this chunk consume, the wave's base state by the chunk's total
        const bool draws_any = pp.n >= 3 && !(pp.flags & (1 | 16));
        const bool integrate = draws_any && !(pp.flags & 6);
        const int my_draws = draws_any ? 8 * (2 * (pp.n - 1) - 1) : 0;
        int incl = my_draws;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int total_draws = __builtin_amdgcn_readlane(incl, 63);
        // the calls of this pass that need a lattice sum: one slot each in the slab's list (order is irrelevant); the rare
        // ones with linearly dependent variables (MVNDFN's grouped limits: the runtime evaluator) fill it from the back
        const bool regular = integrate && pp.closes == (1u << pp.n) - 1u;
        const unsigned long long em = __ballot(regular), cm = __ballot(integrate && !regular);
        unsigned int lbase = 0, cbase = 0;
        if (lane == 0 && em) lbase = atomicAdd(g.count, (unsigned int)__popcll(em));
        if (lane == 0 && cm) cbase = atomicAdd(g.count + 1, (unsigned int)__popcll(cm));
        lbase = (unsigned int)__builtin_amdgcn_readfirstlane((int)lbase);
        cbase = (unsigned int)__builtin_amdgcn_readfirstlane((int)cbase);
        if (mine) {
            if (integrate) {
                const unsigned int id = (unsigned int)(i * total + call);      // record = call of the slab
                double* rec = g.recs + (size_t)id * g.R;
                rec[0] = __longlong_as_double(pack_meta(pp.flags, pp.n, pp.infi, pp.closes));
                rec[1] = (double)id;
                write_slab(pp.n, slab, (regular && ITAL_QMC_FLIP && !g.mixed) ? pp.infi : 0u, rec + 2);      // (subset mode: as MVNDFN has it)
                make_lattice_packed(mrg_jump(d, rng, (uint64_t)(incl - my_draws)), pp.n, slab, rec + g.lat);   // the slab is free now
                if (regular) g.list[lbase + (unsigned int)__popcll(em & ((1ull << lane) - 1ull))] = id;
                else g.list[g.cap - 1u - cbase - (unsigned int)__popcll(cm & ((1ull << lane) - 1ull))] = id;
            }
            double value = pp.value;
            if (!(pp.flags & 1) && (pp.flags & 6)) value = (pp.flags & 2) ? 1.0 : 0.0;
            meta[2 * call] = __longlong_as_double(pack_meta(pp.flags, pp.n, pp.infi, pp.closes));
            meta[2 * call + 1] = value;
        }
        rng = mrg_jump(d, rng, (uint64_t)total_draws);
        wave_sync();
    }
}

// Wide form with the PERFECT user and at most 16 patterns per candidate (fb_mode 0: monte_carlo_num_rel x t sampled patterns,
// BASELINE config 5): every pattern makes one prior call and one call after the simulated update with ALL variables fed
// back.  gen_prep_kernel lets every lane redo the O(n^3) algebra of that update -- W = (Sigma_U + s I)^-1 does not depend on
// the pattern -- in a private LDS scratch that limits a wave to 8 of its 64 lanes at 16 variables (four waves per
// candidate, 0.8 M candidates per second on the whole chip: a tenth of the step, taking a wave slot from the lattice sums
// wherever it runs).  Here ONE wave per candidate
//   A  forms W once, cooperatively (Cholesky by columns with the lanes over the rows, the inverse factor with the lanes over
//      the columns, W = X^T X with the lanes over the pairs: the sums in prepare_call's order);
//   B  lanes 32 + l: the updated call of pattern l is decided from g = W (f - mu): mean' = f - s g, var' = s (1 - s W_aa),
//      early_decision -- n^2 operations on the shared W.  A call this does not decide (noise large against the variances)
//      hands the WHOLE candidate to gen_prep_kernel (redo flag), which runs behind this kernel for the flagged ones only;
//   C  lanes l: the prior call of pattern l through prepare_call as before (COVSRT in the lane's slab), its lattices, its
//      record.
// Stream offsets as in gen_prep_kernel: call c of a candidate starts c x 8 (2 (n - 1) - 1) uniforms behind the candidate's.
__global__ __launch_bounds__(128) void gen_prep_pu_kernel(GArgs a, GPipe g) {
    extern __shared__ double lds_all[];
    const ital_gscore_desc& d = a.d;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t i = (int64_t)blockIdx.x * 2 + wid;
    if (i >= g.slab_n) return;
    const int64_t p = g.slab_lo + i;
    if (!d.alive[p]) return;
    const int n = g.n;                 // plain mode: U = batch so far + candidate, all of them enumerated
    const int ldS = a.ldS;
    double* W0 = lds_all + (size_t)wid * a.wave_doubles;
    double* muU = W0;
    double* SigU = muU + ldS;
    double* Lm = SigU + ldS * ldS;      // lower factor of Sigma_U + s I
    double* Xm = Lm + ldS * ldS;        // its inverse
    double* Wm = Xm + ldS * ldS;        // W, full symmetric
    int* usort = reinterpret_cast<int*>(Wm + ldS * ldS);
    int* ipos = usort + GN;
    double* slabs = Wm + ldS * ldS + (GN + GR + 1) / 2;
    const int row = d.cand[p];
    const int64_t gi = d.row_offset + row;
    const int nE = d.nE;
    const double s = d.noise;
    for (int idx = lane; idx < n * n; idx += 64) {
        const int r = idx / n, c = idx - r * n;
        double v;
        if (r < nE && c < nE) v = d.E_sig[r * d.ldE + c];
        else if (r == c) v = d.s2[row];                         // not clamped (gp.py:254)
        else v = d.C[(int64_t)(r < c ? r : c) * d.ldc + row];
        SigU[r * ldS + c] = v;
    }
    for (int e = lane; e < n; e += 64) muU[e] = e < nE ? d.E_mu[e] : d.mu[row];
    if (lane == 0) {
        int rank = 0;
        for (int e = 0; e < nE; e++) rank += (d.E_idx[e] < gi) ? 1 : 0;
        for (int sidx = 0; sidx < n; sidx++)
            usort[sidx] = sidx < rank ? d.E_sort[sidx] : (sidx == rank ? nE : d.E_sort[sidx - 1]);
        for (int v = 0; v < n; v++) ipos[v] = v < d.n_picks ? d.pick_pos[v] : nE;
    }
    wave_sync();
    // ---- A: W = (Sigma_U + s I)^-1.  Column j of the factor: lane i forms row i's entry (prepare_call's sums, q ascending)
    for (int j = 0; j < n; j++) {
        double v = 0.0;
        if (lane >= j && lane < n) {
            v = SigU[lane * ldS + j] + (lane == j ? s : 0.0);
            for (int q = 0; q < j; q++) v -= Lm[lane * ldS + q] * Lm[j * ldS + q];
        }
        const double dj = sqrt(readlane_f64(v, j));
        if (lane >= j && lane < n) Lm[lane * ldS + j] = lane == j ? dj : v / dj;
        wave_sync();
    }
    if (lane < n) {      // column `lane` of the inverse factor, top to bottom
        const int c = lane;
        Xm[c * ldS + c] = 1.0 / Lm[c * ldS + c];
        for (int r = c + 1; r < n; r++) {
            double sm = 0;
            for (int q = c; q < r; q++) sm += Lm[r * ldS + q] * Xm[q * ldS + c];
            Xm[r * ldS + c] = -sm / Lm[r * ldS + r];
        }
    }
    wave_sync();
    for (int idx = lane; idx < n * (n + 1) / 2; idx += 64) {      // W[r][c] = sum_{k >= r} X[k][c] X[k][r], c <= r
        int r = 0;
        while ((r + 1) * (r + 2) / 2 <= idx) r++;
        const int c = idx - r * (r + 1) / 2;
        double w = 0;
        for (int k = r; k < n; k++) w += Xm[k * ldS + c] * Xm[k * ldS + r];
        Wm[r * ldS + c] = w;
        Wm[c * ldS + r] = w;
    }
    wave_sync();
    // ---- B: the updated calls (lane 32 + l: pattern l), decided from their standardised limits
    const int npat = g.npat;
    const bool upd = lane >= 32 && lane - 32 < npat;
    const int pl_ = upd ? lane - 32 : lane;                     // pattern of this lane
    unsigned relU = 0;
    if (pl_ < npat) {
        const unsigned pat = d.mc_rel > 0 ? d.rel_samples[p * d.mc_rel + pl_] : (unsigned)pl_;
        for (int v = 0; v < n; v++) relU |= ((pat >> (n - 1 - v)) & 1u) << ipos[v];
    }
    int eflag = 1;                                              // (lanes without an updated call count as decided)
    if (upd) {
        const double thr = 37.0 + 9.0 * sqrt((double)(n - 1));
        bool all_full = true, any_empty = false;
        for (int ua = 0; ua < n; ua++) {
            double acc = 0;
            for (int ub = 0; ub < n; ub++) acc += Wm[ua * ldS + ub] * ((((relU >> ub) & 1u) ? 1.0 : -1.0) - muU[ub]);
            const double mean = (((relU >> ua) & 1u) ? 1.0 : -1.0) - s * acc;
            const double var = s * (1.0 - s * Wm[ua * ldS + ua]);
            const double lim = -mean / sqrt(var);
            const double l = ((relU >> ua) & 1u) ? lim : -lim;      // early_decision (gen_common.h)
            if (l > thr) any_empty = true;
            if (!(l < -thr)) all_full = false;
        }
        eflag = any_empty ? 4 : (all_full ? 2 : 0);
    }
    const bool redo = __ballot(eflag == 0) != 0ull;
    if (lane == 0) g.redo[i] = redo ? 1 : 0;
    if (redo) return;                                           // (wave-uniform) gen_prep_kernel takes this candidate
    const int total = g.total;
    double* meta = g.meta + (size_t)i * total * 2;
    if (upd) {
        unsigned infi = 0;                                      // limit types in the call's own order (by data index, ital.py:448)
        for (int q = 0; q < n; q++) infi |= ((relU >> usort[q]) & 1u) << q;
        const int call = 2 * pl_ + 1;
        meta[2 * call] = __longlong_as_double(pack_meta(eflag, n, infi, 0));
        meta[2 * call + 1] = (eflag & 2) ? 1.0 : 0.0;
    }
    // ---- C: the prior calls (lane l: pattern l)
    const bool pri = lane < npat;
    Prep pp;
    pp.n = 0; pp.infi = 0; pp.flags = 16; pp.value = 0; pp.closes = 0; pp.ng = 0; pp.gdraws = 0;
    double* slab = slabs + (size_t)(pri ? lane : 0) * a.stride;
    if (pri) {
        const CallInfo ci = decode_call(d, p, 2 * lane, 2, 1, n, npat);
        pp = prepare_call<false>(prep_opts(d), ci, n, n, ldS, muU, SigU, usort, ipos, false, slab, slab, nullptr);
    }
    const bool integrate = pri && !(pp.flags & (1 | 6 | 16));
    const bool regular = integrate && pp.closes == (1u << pp.n) - 1u;
    const unsigned long long em = __ballot(regular), cm = __ballot(integrate && !regular);
    unsigned int lbase = 0, cbase = 0;
    if (lane == 0 && em) lbase = atomicAdd(g.count, (unsigned int)__popcll(em));
    if (lane == 0 && cm) cbase = atomicAdd(g.count + 1, (unsigned int)__popcll(cm));
    lbase = (unsigned int)__builtin_amdgcn_readfirstlane((int)lbase);
    cbase = (unsigned int)__builtin_amdgcn_readfirstlane((int)cbase);
    if (pri) {
        const int call = 2 * lane;
        if (integrate) {
            // stream position of this call: the candidate's offset in the reference's serial order + `call` calls of this size
            MrgState rng = {d.seed[0], d.seed[1], d.seed[2], d.seed[3], d.seed[4], d.seed[5]};
            const int64_t gpos = d.gpos ? d.gpos[p] : d.pos_offset + p;
            int64_t before = gpos;
            for (int q = 0; q < d.n_dead; q++) before -= (d.dead_pos[q] < gpos) ? 1 : 0;
            uint64_t off = (uint64_t)before * (uint64_t)d.draws_out;
            if (d.draw_off) off = (uint64_t)d.draw_off[p];
            off += (uint64_t)call * (uint64_t)(8 * (2 * (n - 1) - 1));
            const unsigned int id = (unsigned int)(i * total + call);
            double* rec = g.recs + (size_t)id * g.R;
            rec[0] = __longlong_as_double(pack_meta(pp.flags, pp.n, pp.infi, pp.closes));
            rec[1] = (double)id;
            write_slab(pp.n, slab, (regular && ITAL_QMC_FLIP) ? pp.infi : 0u, rec + 2);
            make_lattice_packed(mrg_jump(d, rng, off), pp.n, slab, rec + g.lat);      // the slab is free now
            if (regular) g.list[lbase + (unsigned int)__popcll(em & ((1ull << lane) - 1ull))] = id;
            else g.list[g.cap - 1u - cbase - (unsigned int)__popcll(cm & ((1ull << lane) - 1ull))] = id;
        }
        double value = pp.value;
        if (!(pp.flags & 1) && (pp.flags & 6)) value = (pp.flags & 2) ? 1.0 : 0.0;
        meta[2 * call] = __longlong_as_double(pack_meta(pp.flags, pp.n, pp.infi, pp.closes));
        meta[2 * call + 1] = value;
    }
}

// ------------------------------------------------------------------------------------------------ lattice sums, combine
// The lattice sums of the records a chunk's list names; its length is only known on the device.  T > 0: the regular calls (T
// variables, every row closes its own group) with the compile-time evaluator, one call per wave in a grid that covers the
// capacity of the list (ITAL_GEN_ONE_TRIP); T == 0: the calls with linearly dependent variables, from the back of the
// list, with the runtime evaluator in a small fixed grid of waves that stride over them.
// CF ("reference arithmetic"): the launches of a step with a change-estimation subset.  There the reference weighs
// log(p_U + eps) -- the joint probability over subset + batch + candidate -- with the probability of the enumerated variables
// alone (ital.py:227-275): for a candidate that nearly duplicates a subset member p_U is ~1e-10, EVERY lattice point of it
// runs through intervals [d, 1] with d within 1e-12 of 1, and MVNDFN's own rounding there (the argument d + x (1 - d) of
// Phi^-1 carries 1e-4 relative noise in its distance from 1) is part of the value the reference returns -- with a weight of
// order 1 it reaches the score (golden iris_ce5, candidate 97 next to subset members 96 and 98: 6e-8 at step 2, 1.6e-5 at
// step 4 with the all-upper form).  Everywhere else a probability is weighted by itself and none of this is visible.  So
// these instantiations evaluate MVNDFN as written: lower-bounded variables stay lower-bounded (no ITAL_QMC_FLIP: the
// record holds the factor unsigned), Phi is MVNPHI with its continued fraction beyond |z| = 7.07 (device_math.h WithCF) --
// the arithmetic of the runtime evaluator (T == 0, and the single kernel), at the speed of the compile-time ones.
template <int T, bool CF = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ITAL_GEN_MAIN_WAVES(T), ITAL_GEN_MAIN_WAVES(T)))) void gen_main_kernel(
    GPipe g, const double* __restrict__ vk, unsigned long long* pair_count, int exact) {
    extern __shared__ double lds_all[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lds_lat = g.lat - 2;                                   // slab (factor, limits), then the unpacked lattices
    // (T == 0: one wave per workgroup, which also holds the chains' conditioned values in LDS: qmc_eval_lds)
    constexpr int YLD = T >= 7 ? ITAL_BIG_YLDS(T) * ITAL_GEN_BIG_NCB(T) * 64 : 0;   // conditioned values of the last stages (qmc_common.h)
    constexpr int TQ = T >= 7 ? 64 * ITAL_GEN_BIG_NCB(T) : ITAL_GEN_TAILQ;           // tail queue: one slot per chain and lane
    double* rec = lds_all + (size_t)wid * (lds_lat + 16 * (g.n - 1) + TQ + YLD);
    double* tailq = rec + lds_lat + 16 * (g.n - 1);
    const unsigned int count = g.count[T > 0 ? 0 : 1];
    const unsigned int nwaves = gridDim.x * (blockDim.x >> 6);
    unsigned long long pairs = 0;
    auto integrate = [&](unsigned int e) {
        const unsigned int r = g.list[T > 0 ? e : g.cap - 1u - e];
        const double* src = g.recs + (size_t)r * g.R;
        const long long m = __double_as_longlong(uniform_f64(src[0]));
        const unsigned int id = (unsigned int)uniform_f64(src[1]);
        const int n = (int)((m >> 8) & 0xff);
        if (T > 0 && n != T) return;          // (wave-uniform; subset mode: another launch takes the records of n variables)
        const unsigned infi = (unsigned)((m >> 16) & 0xffffffu);
        const int ns = n * (n + 1) / 2 + n, ndim = n - 1;
        // the compile-time evaluators take every variable as bounded above (ITAL_QMC_FLIP): the record holds the factor and
        // the limits with the signs already in place, the shifts of a negated variable move by 1/2 here
        const unsigned fl = (T > 0 && ITAL_QMC_FLIP && !CF) ? infi : 0u;
        for (int q = lane; q < ns; q += 64) rec[q] = src[2 + q];
        {   // unpack the lattices: generator = vk[n][index], shift = integer * 1/(m1 + 1) exactly as MVNUNI forms it
            const unsigned int* shifts = reinterpret_cast<const unsigned int*>(src + g.lat);
            const unsigned long long* perm = reinterpret_cast<const unsigned long long*>(src + g.lat + 4 * ndim);
            for (int q = lane; q < 8 * ndim; q += 64) {
                const int sft = q / ndim, j = q - sft * ndim;
                rec[lds_lat + q] = vk[n * GN + (int)((perm[sft] >> (4 * j)) & 15ull)];
                rec[lds_lat + 8 * ndim + q] = (double)shifts[q] * MRG_INVMP1 + (((fl >> j) & 1u) ? 0.5 : 0.0);
            }
        }
        constexpr bool FL = T > 0 && ITAL_QMC_FLIP != 0 && !CF;
        wave_sync();
        double value;
        if (T >= 7) {
            constexpr int TB = T >= 7 ? T : 7, NDIMB = TB - 1;
#if ITAL_GEN_BIG_HOTK
            // (the far-tail branch of the CF instantiations takes a few registers: two exp coefficients fewer in registers there)
            constexpr int KN = ITAL_GEN_BIG_KEN(TB) - (CF && ITAL_GEN_BIG_KEN(TB) >= 2 ? 2 : 0);
            typename std::conditional<CF, WithCF<HotKEn<KN>>, ITAL_GEN_BIG_COEF(TB)>::type kk;      // exp coefficients as vector-register operands (device_math.h), as in the perfect-user kernel
            kk.load();
            value = wave_sum(qmc_lane_sum_big<TB, ITAL_GEN_BIG_NCB(TB), decltype(kk), FL, ITAL_BIG_YLDS(TB), ITAL_BIG_GROUP(TB)>(
                        rec + lds_lat, rec, infi, tailq, lane, kk, tailq + TQ)) /
                    (16.0 * P_TAB[(NDIMB < 10 ? NDIMB : 10) - 1]);
#else
            value = wave_sum(qmc_lane_sum_big<TB, ITAL_GEN_BIG_NCB(TB), LitK, FL, ITAL_BIG_YLDS(TB), ITAL_BIG_GROUP(TB)>(
                        rec + lds_lat, rec, infi, tailq, lane, LitK(), tailq + TQ)) /
                    (16.0 * P_TAB[(NDIMB < 10 ? NDIMB : 10) - 1]);
#endif
        } else if (T > 0) {
            constexpr int TF = T > 0 && T < 7 ? T : 3;
            value = qmc_eval_fixed_inl<TF, ITAL_GEN_FIXED_NH(TF), FL, CF>(rec, infi, rec + lds_lat, lane, tailq);
        }
        else value = qmc_eval_lds(n, rec, infi, (unsigned)((m >> 40) & 0xffffffu), rec + lds_lat, lane, tailq, tailq + TQ);
        if (lane == 0) {
            g.meta[(size_t)id * 2 + 1] = value;
            // label_estimation 'optimistic' / 'pessimistic' compare terms for exact equality: a sum this close to 0 or 1 is
            // formed again in the reference's own order (gen_exact_kernel, right after this launch)
            if (exact && (value > 1.0 - EXACT_BAND || value < EXACT_BAND))
                g.recs[(size_t)r * g.R] = __longlong_as_double(m | GEN_FLAG_EXACT);
        }
        pairs += 16ull * P_TAB[(n - 1 < 10 ? n - 1 : 10) - 1] * (n - 1);
        wave_sync();
    };
    if (ITAL_GEN_ONE_TRIP(T)) {
        // one call per wave, the grid covers the capacity of the list: no loop around the evaluator, whose register
        // allocation then is the one of the perfect-user kernel
        const unsigned int e = blockIdx.x * 4 + wid;
        if (e < count) integrate(e);
        // every regular call of this launch has the same size: one atomic for the whole list instead of one per wave
        // (185 k waves adding to one address cost 1.7 ms per noisy-user step)
        constexpr int ND = T > 1 ? T - 1 : 1;
        if (!g.mixed)      // (mixed dimensions: every wave reports what it integrated -- instrumentation, off the timed path)
            pairs = (blockIdx.x == 0 && wid == 0) ? (unsigned long long)count * (16ull * P_TAB[(ND < 10 ? ND : 10) - 1] * ND) : 0ull;
    } else {
        for (unsigned int e = blockIdx.x * (blockDim.x >> 6) + wid; e < count; e += nwaves) integrate(e);
    }
    if (lane == 0 && pair_count && pairs) atomicAdd(pair_count, pairs);
}

// The flagged records of a chunk again, in the reference's summation order (qmc_exact.h): wave per record; all but a
// handful leave at once.  Launched only with label_estimation 'optimistic' / 'pessimistic'; any dimension the pipeline takes
// (round 6: the lattice points in blocks, qmc_exact_lds -- until then 16 P values in LDS limited this to 8 variables).
__global__ __launch_bounds__(64) void gen_exact_kernel(GPipe g, const double* __restrict__ vk, unsigned int nrec) {
    extern __shared__ double lds_all[];
    const int lane = threadIdx.x;
    // the block index runs over the chunk's list: regular calls from the front, the others from the back (records outside
    // the list are leftovers of an earlier chunk)
    // (a bounded grid strides over the list: with most calls decided before they reach a record, one workgroup per slot of the
    // chunk's capacity -- up to 2^20, all but a handful leaving at once -- was launched for nothing)
    for (unsigned int e = blockIdx.x; e < nrec; e += gridDim.x) {
    if (e >= g.count[0] && e < nrec - g.count[1]) continue;
    const unsigned int r = g.list[e];
    const double* src = g.recs + (size_t)r * g.R;
    const long long m = __double_as_longlong(uniform_f64(src[0]));
    if (!(m & GEN_FLAG_EXACT)) continue;
    const unsigned int id = (unsigned int)uniform_f64(src[1]);
    const int n = (int)((m >> 8) & 0xff);
    const unsigned infi = (unsigned)((m >> 16) & 0xffffffu), closes = (unsigned)((m >> 40) & 0xffffffu);
    const int ncov = n * (n + 1) / 2, ndim = n - 1;
    double* slab = lds_all;
    double* lat = slab + ncov + n;
    double* tailq = lat + 16 * ndim;
    double* vals = tailq + 128;
    double* yl = vals + 128;                 // conditioned values of the two chains per lane: [2][ndim][64]
    // a regular call's record holds the variables bounded below negated (write_slab): undo
    const unsigned fl = (ITAL_QMC_FLIP && !g.mixed && closes == (1u << n) - 1u) ? infi : 0u;
    for (int q = lane; q < ncov; q += 64) {
        int row = 0;
        while ((row + 1) * (row + 2) / 2 <= q) row++;
        const int col = q - row * (row + 1) / 2;
        slab[q] = (((fl >> row) ^ (fl >> col)) & 1u) ? -src[2 + q] : src[2 + q];
    }
    for (int q = lane; q < n; q += 64) slab[ncov + q] = ((fl >> q) & 1u) ? -src[2 + ncov + q] : src[2 + ncov + q];
    {
        const unsigned int* shifts = reinterpret_cast<const unsigned int*>(src + g.lat);
        const unsigned long long* perm = reinterpret_cast<const unsigned long long*>(src + g.lat + 4 * ndim);
        for (int q = lane; q < 8 * ndim; q += 64) {
            const int sft = q / ndim, j = q - sft * ndim;
            lat[q] = vk[n * GN + (int)((perm[sft] >> (4 * j)) & 15ull)];
            lat[8 * ndim + q] = (double)shifts[q] * MRG_INVMP1;
        }
    }
    wave_sync();
    const double value = qmc_exact_lds(n, slab, infi, closes, lat, lane, tailq, yl, ndim, vals);
    if (lane == 0) g.meta[(size_t)id * 2 + 1] = value;
    wave_sync();
    }
}

// Wave per candidate: the lanes form the terms of 64 calls at a time, lane 0's order-preserving fold adds them up exactly
// as the reference's loop does (ital.py:207-222).
__global__ __launch_bounds__(256) void gen_combine_kernel(ital_gscore_desc d, GPipe g) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= g.slab_n) return;
    const int64_t p = g.slab_lo + i;
    if (!d.alive[p]) return;
    const int nr = d.n_picks + 1;
    const int npat = g.npat, cpp = g.cpp, total = g.total;
    const bool entropy = d.fb_mode == 3;
    const double* meta = g.meta + (size_t)i * total * 2;
    double mi = 0.0;
    for (int c0 = 0; c0 < total; c0 += 64) {
        const int call = c0 + lane;
        // kind of term this lane contributes: 0 none, 1 updated-call term (mean / sampled), 2 entropy term, 3 single entropy
        int kind = 0;
        double term = 0.0;
        if (call < total) {
            const int fl = (int)(__double_as_longlong(meta[2 * call]) & 0xff);
            if (!(fl & 16)) {
                const double value = meta[2 * call + 1];
                const CallInfo ci = decode_call(d, p, call, cpp, g.npre, nr, npat);
                if (entropy) {
                    if (nr == 1) {
                        if (ci.pat == 0) {
                            const double q = fmax(1e-8, fmin(1.0 - 1e-8, value));
                            term = q * log(q) + (1.0 - q) * log(1.0 - q);
                            kind = 3;
                        }
                    } else if (value > 1e-12) {
                        term = value * log(value);
                        kind = 2;
                    }
                } else if (ci.kind == K_UPDATED) {
                    // the pattern's prior probability: of the enumerated variables alone (the weight) and -- with a
                    // change-estimation subset -- of all of U (the reference of the logarithm), ital.py:227-275
                    const int base = (call / cpp) * cpp;
                    const double pr = meta[2 * base + 1];
                    const double prl = g.npre == 2 ? meta[2 * (base + 1) + 1] : pr;
                    const double cur = (log(value + d.eps) - log(prl + d.eps)) * ci.weight;
                    const bool lm = d.label_mode != 0 && g.npre == 1;          // (label estimates: plain mode only)
                    term = (lm || d.mc_rel > 0) ? cur : cur * pr;   // sampled patterns are not weighted
                    kind = 1;
                }
            }
        }
        for (int l = 0; l < 64 && c0 + l < total; l++) {
            const int k_l = __builtin_amdgcn_readlane(kind, l);
            if (k_l == 0) continue;
            const double t_l = readlane_f64(term, l);
            if (k_l == 3) mi = t_l;
            else if (k_l == 2) mi += t_l;
            else if (d.label_mode == 1 && g.npre == 1) { if (t_l > mi) mi = t_l; }
            else if (d.label_mode == 2 && g.npre == 1) { if (mi == 0 || t_l < mi) mi = t_l; }
            else mi += t_l;
        }
    }
    if (d.mc_rel > 0) mi /= d.mc_rel;   // ital.py:221-222
    if (entropy) mi = -mi;
    if (lane == 0) d.mi[p] = mi;
}

}  // namespace ital

using namespace ital;

static int fs_doubles(int nr) { return nr * nr + 2 * nr; }

// LDS doubles per wave of gen_main_kernel<n> for its Phi^-1 tail queue and the conditioned values it keeps there
// (ITAL_BIG_YLDS, qmc_common.h)
static int gen_main_yld(int n) {
    switch (n) {
#define ITAL_YLD_CASE(T_) case T_: return 64 * ITAL_GEN_BIG_NCB(T_) + ITAL_BIG_YLDS(T_) * ITAL_GEN_BIG_NCB(T_) * 64;
        ITAL_YLD_CASE(7) ITAL_YLD_CASE(8) ITAL_YLD_CASE(9) ITAL_YLD_CASE(10) ITAL_YLD_CASE(11) ITAL_YLD_CASE(12)
        ITAL_YLD_CASE(13) ITAL_YLD_CASE(14) ITAL_YLD_CASE(15) ITAL_YLD_CASE(16)
#undef ITAL_YLD_CASE
    }
    return ITAL_GEN_TAILQ;
}

// Streams and events of the pipeline (one set per device of the process, created on first use).
struct PipeStreams {
    hipStream_t prep, main;
    hipEvent_t start, built[2], summed[2], combined;
    // wide form: slabs launched so far on this device (buffer = parity) and whether a buffer's `summed` event has ever been
    // recorded -- kept across calls: with ital_gscore_desc.defer_join the next call's first slab is prepared while this
    // call's last one is still being integrated
    unsigned long long wide_slabs;
    bool summed_valid[2];
    bool pending;            // kernels may still run on prep / main that the caller's stream has not been joined with
    unsigned int* host_count;   // page-locked landing word: the length of list U after a slab's verdicts (fast form)
};

static PipeStreams* pipe_streams() {
    static PipeStreams sets[16];
    static bool made[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!made[dev]) {
        PipeStreams& p = sets[dev];
        bool ok = hipStreamCreateWithFlags(&p.prep, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&p.main, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&p.start, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&p.combined, hipEventDisableTiming) == hipSuccess;
        for (int q = 0; q < 2 && ok; q++)
            ok = hipEventCreateWithFlags(&p.built[q], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&p.summed[q], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&p.host_count), 64, hipHostMallocDefault) == hipSuccess;
        if (!ok) return nullptr;
        p.wide_slabs = 0;
        p.summed_valid[0] = p.summed_valid[1] = false;
        p.pending = false;
        made[dev] = true;
    }
    return &sets[dev];
}

namespace {

// Error exit once kernels may be running on the internal streams: the caller's stream is made to wait for both of them
// before the code is returned -- the caller may free or reuse the workspace as soon as ITS stream has drained.
int pipe_bail(PipeStreams* ps, hipStream_t stream, int rc) {
    (void)hipEventRecord(ps->built[0], ps->prep);
    (void)hipStreamWaitEvent(ps->main, ps->built[0], 0);
    (void)hipEventRecord(ps->combined, ps->main);
    (void)hipStreamWaitEvent(stream, ps->combined, 0);
    ps->pending = false;
    return rc;
}

struct PipePlan {
    bool ok;            // the step is the pipeline's
    bool fast;          // 3 .. 6 variables
    bool sub;           // change-estimation subset (wide form whatever the dimension)
    int npre;
    int n, nr;
    int64_t total;      // calls per candidate
    int npat, cpp;
    int R, lat;
    int64_t per_cand;   // doubles per candidate of a slab (fast: state, meta, list U; wide: meta, records, list -- per buffer)
    int64_t chunk_max;  // fast: records per chunk buffer at most
};

PipePlan pipe_plan(const ital_gscore_desc* d) {
    PipePlan pl = {};
    const int nr = d->n_picks + 1, n = d->nE + 1;      // n: the most variables a call of this step has (U = E + candidate)
    const bool subset = d->subset_mode != 0;
    if (n < 1 || n > ITAL_GEN_TFIX_MAX || nr < 1 || (subset ? nr > n : nr != n)) return pl;
    if (subset && (!ITAL_GEN_PIPE_SUBSET || d->fb_mode == 3 || n > ITAL_GEN_SUB_MAX)) return pl;
    if (d->clip_cov > 0 && d->clip_cov < 1 && n > 5) return pl;
    const double npat = d->mc_rel > 0 ? (double)d->mc_rel : pow(2.0, nr);
    const double nfb = d->fb_mode == 3 ? 0.0 : d->fb_mode == 0 ? 1.0 : (d->mc_fb > 0 ? (double)d->mc_fb : (d->fb_mode == 1 ? pow(2.0, nr) : pow(3.0, nr) - 1));
    const int npre = subset ? 2 : 1;                   // prior calls per pattern (ital.py:227-275: enumerated variables, then all of U)
    if (npat * (npre + nfb) > (double)ITAL_GENERIC_MAX_CALLS) return pl;
    pl.ok = true;
    pl.sub = subset;
    pl.npre = npre;
    pl.fast = !subset && n <= 6;                       // (a subset always takes the wide form: its calls differ in dimension)
    pl.n = n; pl.nr = nr;
    pl.npat = (int)npat; pl.cpp = npre + (int)nfb;
    pl.total = (int64_t)pl.npat * pl.cpp;
    pl.lat = 2 + n * (n + 1) / 2 + n;
    pl.R = pl.lat + 4 * (n - 1) + 8;
    if (n <= 2 && !subset) {           // closed forms: the verdicts are the values
        pl.per_cand = pl.total * 2;
        pl.chunk_max = 0;
    } else if (pl.fast) {
        pl.per_cand = 3 + pl.total * 2 + (pl.total + 1) / 2;
        pl.chunk_max = (int64_t)1 << 20;
    } else {
        pl.per_cand = pl.total * (2 + (int64_t)pl.R) + (pl.total + 1) / 2 + 1;      // meta, records, list, the redo flag
    }
    return pl;
}

constexpr int64_t HDR = 64;  // doubles in front of the buffers: the counters -- the length of list U, then one (front, back) pair of
                             // list lengths per chunk of a slab (GEN_PAIRS of them: cleared by ONE fill per slab; chunks beyond
                             // re-use the last two pairs behind a fill of their own)
constexpr int GEN_PAIRS = (int)(HDR * 2 - 2) / 2;

}  // namespace

// Doubles of workspace with which ital_score_generic runs this step's candidates through the pipeline in one slab (0: the
// step is not the pipeline's); a smaller workspace means more slabs (at least one candidate must fit).
extern "C" int64_t ital_score_generic_workspace(const ital_gscore_desc* d) {
    if (!d || d->n_cand <= 0) return 0;
    const PipePlan pl = pipe_plan(d);
    if (!pl.ok) return 0;
    if (!pl.fast && (pl.sub || pl.n > 2)) return 2 * (1 + pl.per_cand * d->n_cand);
    if (pl.n <= 2) return HDR + pl.per_cand * d->n_cand;
    int64_t ch = pl.chunk_max;
    while (ch > 4096 && ch / 2 >= d->n_cand * pl.total) ch >>= 1;
    return HDR + pl.per_cand * d->n_cand + 2 * (ch * pl.R + ch / 2 + 1);
}

int ital_gen_pipeline(const ital_gscore_desc* d, hipStream_t stream) {
    const PipePlan pl = pipe_plan(d);
    if (!pl.ok || !d->work) return 1;
    const int n = pl.n;
    GPipe g = {};
    g.total = (int)pl.total; g.npat = pl.npat; g.cpp = pl.cpp; g.n = n; g.R = pl.R; g.lat = pl.lat;
    g.npre = pl.npre; g.mixed = pl.sub ? 1 : 0;
    if (pl.fast || (n <= 2 && !pl.sub)) (void)ital_score_generic_join(stream);     // (a deferred call still pending: these forms share their buffers between calls)
    if (n <= 2 && !pl.sub) {
        // ---- one or two variables: closed forms and combine on the caller's stream, slabs of the workspace
        int64_t S = (d->work_doubles - HDR) / pl.per_cand;
        if (S < 1) return 1;
        if (S > d->n_cand) S = d->n_cand;
        g.meta = d->work + HDR;
        for (int64_t lo = 0; lo < d->n_cand; lo += S) {
            g.slab_lo = lo;
            g.slab_n = d->n_cand - lo < S ? d->n_cand - lo : S;
            const unsigned threads = pl.total >= 256 ? 256u : (unsigned)((pl.total + 63) / 64 * 64);
            if (n == 1) ITAL_LAUNCH(gen_closed_kernel<1>, dim3((unsigned)g.slab_n), dim3(threads), 0, stream, *d, g);
            else ITAL_LAUNCH(gen_closed_kernel<2>, dim3((unsigned)g.slab_n), dim3(threads), 0, stream, *d, g);
            ITAL_LAUNCH(gen_combine_kernel, dim3((unsigned)((g.slab_n + 3) / 4)), dim3(256), 0, stream, *d, g);
            const int rc = ital_check_launch("ital_score_generic(closed forms)");
            if (rc) return rc;
        }
        return 0;
    }
    PipeStreams* ps = pipe_streams();
    if (!ps) return ital_fail(-12, "ital_score_generic: cannot create the pipeline's streams");

    // dimensions of the calls that may need a lattice sum: n in plain mode; with a subset the enumerated variables alone (nr),
    // E (nE = n - 1: a candidate that is a member of E) and E + candidate (n) -- one gen_main_kernel<T> launch each
    int dims[3] = {n, 0, 0};
    int ndims = 1;
    if (pl.sub) {
        ndims = 0;
        const int cands[3] = {pl.nr, n - 1, n};
        for (int q = 0; q < 3; q++) {
            bool dup = cands[q] < 3;
            for (int r_ = 0; r_ < ndims; r_++) dup = dup || dims[r_] == cands[q];
            if (!dup) dims[ndims++] = cands[q];
        }
    }
    // LDS of a lattice-sum workgroup (4 waves): slab + lattices sized for the step's largest call, tail queue and conditioned
    // values as the instantiation needs them (gen_main_yld)
    size_t lds_m = 0;
    for (int q = 0; q < ndims; q++)
        lds_m = std::max(lds_m, (size_t)4 * (pl.lat - 2 + 16 * (n - 1) + gen_main_yld(dims[q])) * sizeof(double));
    const size_t lds_m0 = (size_t)(pl.lat - 2 + 16 * (n - 1) + ITAL_GEN_TAILQ + 2 * (GN - 1) * 64) * sizeof(double);   // one wave
    // label_estimation 'optimistic' / 'pessimistic' (plain mode only): sums that decide an exact comparison are formed again in
    // the reference's order
    const int exact = (d->label_mode != 0 && d->fb_mode != 3 && !pl.sub) ? 1 : 0;
    const size_t lds_x = (size_t)(n * (n + 1) / 2 + n + 16 * (n - 1) + 128 + 128 + 2 * (n - 1) * 64) * sizeof(double);      // <= 21 KB
// (subset mode: the instantiations with MVNPHI's far-tail branch, up to ITAL_GEN_SUB_MAX variables -- pipe_plan sends larger
// subsets to the single kernel; TS_ keeps the dimensions beyond from being instantiated at all)
#define ITAL_GEN_MAIN(T_) case T_: {                                                                                           \
        constexpr int TS_ = T_ <= ITAL_GEN_SUB_MAX ? T_ : ITAL_GEN_SUB_MAX;                                                    \
        if (pl.sub) ITAL_LAUNCH((gen_main_kernel<TS_, true>), dim3(ITAL_GEN_ONE_TRIP(T_) ? (g.cap + 3) / 4 : 768), dim3(256), lds_m, ps->main, g, d->vk, d->pair_count, exact); \
        else ITAL_LAUNCH((gen_main_kernel<T_, false>), dim3(ITAL_GEN_ONE_TRIP(T_) ? (g.cap + 3) / 4 : 768), dim3(256), lds_m, ps->main, g, d->vk, d->pair_count, exact); \
        } break;
#define ITAL_GEN_MAIN_LDS(T_) case T_: { static ItalLdsFlags f_, fc_;                                                         \
        constexpr int TS_ = T_ <= ITAL_GEN_SUB_MAX ? T_ : ITAL_GEN_SUB_MAX;                                                    \
        rc_lds = pl.sub ? ital_raise_lds_limit(reinterpret_cast<const void*>(&gen_main_kernel<TS_, true>), (int)lds_m, fc_, "ital_score_generic") \
                        : ital_raise_lds_limit(reinterpret_cast<const void*>(&gen_main_kernel<T_, false>), (int)lds_m, f_, "ital_score_generic"); } break;
    if (lds_m > 48 * 1024)         // conditioned values of the last stages in LDS (ITAL_BIG_YLDS) can take the workgroup beyond the default limit
      for (int q = 0; q < ndims; q++) {
        int rc_lds = 0;
        switch (dims[q]) {
            ITAL_GEN_MAIN_LDS(7) ITAL_GEN_MAIN_LDS(8) ITAL_GEN_MAIN_LDS(9) ITAL_GEN_MAIN_LDS(10) ITAL_GEN_MAIN_LDS(11) ITAL_GEN_MAIN_LDS(12)
            ITAL_GEN_MAIN_LDS(13) ITAL_GEN_MAIN_LDS(14) ITAL_GEN_MAIN_LDS(15) ITAL_GEN_MAIN_LDS(16)
        }
        if (rc_lds) return rc_lds;
    }
#undef ITAL_GEN_MAIN_LDS
#define ITAL_GEN_MAINS()                                                                                                      \
    for (int q_ = 0; q_ < ndims; q_++)                                                                                        \
    switch (dims[q_]) {                                                                                                       \
        ITAL_GEN_MAIN(3) ITAL_GEN_MAIN(4) ITAL_GEN_MAIN(5) ITAL_GEN_MAIN(6) ITAL_GEN_MAIN(7) ITAL_GEN_MAIN(8) ITAL_GEN_MAIN(9)  \
        ITAL_GEN_MAIN(10) ITAL_GEN_MAIN(11) ITAL_GEN_MAIN(12) ITAL_GEN_MAIN(13) ITAL_GEN_MAIN(14) ITAL_GEN_MAIN(15)            \
        ITAL_GEN_MAIN(16)                                                                                                     \
    }                                                                                                                         \
    ITAL_LAUNCH((gen_main_kernel<0, false>), dim3(256), dim3(64), lds_m0, ps->main, g, d->vk, d->pair_count, exact);             \
    if (exact) ITAL_LAUNCH(gen_exact_kernel, dim3(g.cap < 4096u ? g.cap : 4096u), dim3(64), lds_x, ps->main, g, d->vk, g.cap)

    if (pl.fast) {
        // ---- workspace: counters | per slab: generator states, meta, list U | two chunk buffers: records, list
        int64_t ch = pl.chunk_max;
        while (ch > 4096 && ch / 2 >= d->n_cand * pl.total) ch >>= 1;
        while (ch > 64 && HDR + pl.per_cand + 2 * (ch * pl.R + ch / 2 + 1) > d->work_doubles) ch >>= 1;
        const int64_t chunk_doubles = ch * pl.R + ch / 2 + 1;
        int64_t S = (d->work_doubles - HDR - 2 * chunk_doubles) / pl.per_cand;
        if (S < 1) return 1;                              // not even one candidate: the single kernel takes the step
        if (S > d->n_cand) S = d->n_cand;
        while (S * pl.total >= ((int64_t)1 << 31)) S >>= 1;     // 32-bit call ids
        unsigned int* counters = reinterpret_cast<unsigned int*>(d->work);
        double* base = d->work + HDR;
        g.cstate = reinterpret_cast<int*>(base);
        g.meta = base + 3 * S;
        g.listU = reinterpret_cast<unsigned int*>(g.meta + S * pl.total * 2);
        double* chunk0 = base + pl.per_cand * S;
        g.countU = counters;
        g.cap = (unsigned int)ch;
        const int stride_b = (n * (n + 1) / 2 + 2 * n) | 1;
        const size_t lds_b = (size_t)256 * stride_b * sizeof(double);
        if (lds_b > 48 * 1024) {       // six variables: 66 KB of per-thread slabs per workgroup (before anything is enqueued)
            static ItalLdsFlags build_flags;
            if (const int rc = ital_raise_lds_limit(reinterpret_cast<const void*>(&gen_build_kernel<6>), (int)lds_b, build_flags,
                                                    "ital_score_generic"))
                return rc;
        }
        if (hipEventRecord(ps->start, stream) != hipSuccess || hipStreamWaitEvent(ps->prep, ps->start, 0) != hipSuccess ||
            hipStreamWaitEvent(ps->main, ps->start, 0) != hipSuccess)
            return ital_fail(-5, "ital_score_generic: stream synchronisation failed");
        int nbuf = 0;                                     // chunk launches so far (buffer = parity)
        for (int64_t lo = 0; lo < d->n_cand; lo += S) {
            g.slab_lo = lo;
            g.slab_n = d->n_cand - lo < S ? d->n_cand - lo : S;
            if (lo > 0) (void)hipStreamWaitEvent(ps->prep, ps->combined, 0);      // meta / list U are free again
            // (every counter of the slab in one fill; no chunk of the slab before is still being built or summed: `combined`
            // above, or -- first slab -- the caller's stream / the join at the top)
            (void)hipMemsetAsync(counters, 0, (size_t)HDR * sizeof(double), ps->prep);
            GSeed sd = {*d, g.slab_lo, g.slab_n, g.cstate};
            ITAL_LAUNCH(gen_seed_kernel, dim3((unsigned)((g.slab_n + 255) / 256)), dim3(256), 0, ps->prep, sd);
            // workgroup width of the verdict kernel: the one that leaves the fewest lanes idle over a candidate's calls
            unsigned vthreads = 256;
            {
                double best = 0;
                for (unsigned w = 256; w >= 64; w -= 64) {
                    const double eff = (double)pl.total / (double)(((pl.total + w - 1) / w) * w);
                    if (eff > best + 1e-9) { best = eff; vthreads = w; }
                }
            }
            const dim3 vgrid((unsigned)g.slab_n);
            switch (n) {
                case 3: ITAL_LAUNCH(gen_verdict_kernel<3>, vgrid, dim3(vthreads), 0, ps->prep, *d, g); break;
                case 4: ITAL_LAUNCH(gen_verdict_kernel<4>, vgrid, dim3(vthreads), 0, ps->prep, *d, g); break;
                case 5: ITAL_LAUNCH(gen_verdict_kernel<5>, vgrid, dim3(vthreads), 0, ps->prep, *d, g); break;
                case 6: ITAL_LAUNCH(gen_verdict_kernel<6>, vgrid, dim3(vthreads), 0, ps->prep, *d, g); break;
            }
            // How many calls the verdicts left undecided is only known on the device.  Until round 5 the chunk iterations were
            // launched for the worst case (every call undecided: 12 iterations of build / lattice sums / fill per noisy-user
            // step at t = 4, of which 3 found work -- the rest ran as empty grids of up to 2^18 workgroups, ~0.1 ms each).  The
            // length of list U now comes back through a 4-byte copy behind the verdict kernel: one host wait per slab (the
            // lattice sums of the slab before keep the GPU busy meanwhile), then exactly the chunks that hold entries.
            int64_t nchunks = (g.slab_n * pl.total + ch - 1) / ch;
            if (ITAL_GEN_COUNT_SYNC) {
                if (hipMemcpyAsync(ps->host_count, g.countU, sizeof(unsigned int), hipMemcpyDeviceToHost, ps->prep) != hipSuccess ||
                    hipStreamSynchronize(ps->prep) != hipSuccess)
                    return pipe_bail(ps, stream, ital_fail(-5, "ital_score_generic: download of the undecided-call count failed"));
                const int64_t n_und = (int64_t)*ps->host_count;
                nchunks = (n_und + ch - 1) / ch;
            }
            for (int64_t c = 0; c < nchunks; c++, nbuf++) {
                const int buf = nbuf & 1;
                double* cb = chunk0 + (size_t)buf * chunk_doubles;
                g.recs = cb;
                g.list = reinterpret_cast<unsigned int*>(cb + ch * pl.R);
                // the chunk's list lengths: a pair of its own (cleared with the slab's fill) while they last
                const bool own_pair = c < GEN_PAIRS - 2;
                g.count = counters + 2 + 2 * (own_pair ? (int)c : GEN_PAIRS - 2 + buf);
                g.chunk_lo = (unsigned int)(c * ch);
                if (nbuf >= 2) (void)hipStreamWaitEvent(ps->prep, ps->summed[buf], 0);      // the buffer is free again
                if (!own_pair) (void)hipMemsetAsync(g.count, 0, 2 * sizeof(unsigned int), ps->prep);
                const dim3 bgrid((unsigned)((ch + 255) / 256));
                switch (n) {
                    case 3: ITAL_LAUNCH(gen_build_kernel<3>, bgrid, dim3(256), lds_b, ps->prep, *d, g); break;
                    case 4: ITAL_LAUNCH(gen_build_kernel<4>, bgrid, dim3(256), lds_b, ps->prep, *d, g); break;
                    case 5: ITAL_LAUNCH(gen_build_kernel<5>, bgrid, dim3(256), lds_b, ps->prep, *d, g); break;
                    case 6: ITAL_LAUNCH(gen_build_kernel<6>, bgrid, dim3(256), lds_b, ps->prep, *d, g); break;
                }
                (void)hipEventRecord(ps->built[buf], ps->prep);
                (void)hipStreamWaitEvent(ps->main, ps->built[buf], 0);
                ITAL_GEN_MAINS();
                (void)hipEventRecord(ps->summed[buf], ps->main);
            }
            ITAL_LAUNCH(gen_combine_kernel, dim3((unsigned)((g.slab_n + 3) / 4)), dim3(256), 0, ps->main, *d, g);
            (void)hipEventRecord(ps->combined, ps->main);
            const int rc = ital_check_launch("ital_score_generic(pipeline)");
            if (rc) return pipe_bail(ps, stream, rc);
        }
        (void)hipStreamWaitEvent(stream, ps->combined, 0);
        return 0;
    }

    // ---- wide form: two buffers of (counters, meta, records, list), a slab of candidates each
    const int64_t half = d->work_doubles / 2 - 1;
    if (half < pl.per_cand) return 1;
    int64_t S = half / pl.per_cand;
    if (S > d->n_cand) S = d->n_cand;
    while (S * pl.total >= ((int64_t)1 << 31)) S >>= 1;     // 32-bit list entries
    GArgs ap;
    ap.d = *d;
    const int slab = n * (n + 1) / 2 + 2 * n;
    const int stride_p = (slab + fs_doubles(pl.nr)) | 1;
    int chunk_p = 64;
    while (chunk_p > 4 && chunk_p * stride_p > 4096) chunk_p >>= 1;   // <= 32 KB of call slabs per wave
    ap.chunk = chunk_p;
    ap.stride = stride_p;
    ap.slab = slab;
    ap.lat = 0;
    ap.master = 0;
    ap.yl = 0;
    ap.ldS = n;
    ap.wave_doubles = n + n * n + (GN + GR + 1) / 2 + chunk_p * stride_p;
    const size_t lds_p = (size_t)2 * ap.wave_doubles * sizeof(double);
    if (lds_p > 160 * 1024) return ital_fail(-12, "ital_score_generic: LDS budget exceeded");
    // a slab is a few hundred candidates: one wave each would leave most of the chip idle during the preparation
    const int npass = (int)((pl.total + chunk_p - 1) / chunk_p);
    g.nsplit = (d->mc_fb > 0 || pl.sub) ? 1 : (npass < ITAL_GEN_PREP_SPLIT ? (npass < 1 ? 1 : npass) : ITAL_GEN_PREP_SPLIT);
    static ItalLdsFlags prep_flags;
    if (const int rc = ital_raise_lds_limit(reinterpret_cast<const void*>(&gen_prep_kernel), 160 * 1024, prep_flags, "ital_score_generic"))
        return rc;
    // perfect user with a few patterns per candidate (sampled patterns): the cooperative preparation, gen_prep_kernel behind it
    // for the candidates it flags
    const bool pu = ITAL_GEN_PREP_PU && !pl.sub && d->fb_mode == 0 && d->mc_fb == 0 && pl.cpp == 2 && pl.npat <= 16;
    GArgs apu = ap;
    size_t lds_pu = 0;
    if (pu) {
        apu.stride = slab | 1;
        apu.wave_doubles = n + 4 * n * n + (GN + GR + 1) / 2 + pl.npat * apu.stride;
        lds_pu = (size_t)2 * apu.wave_doubles * sizeof(double);
        static ItalLdsFlags pu_flags;
        if (lds_pu > 48 * 1024)
            if (const int rc = ital_raise_lds_limit(reinterpret_cast<const void*>(&gen_prep_pu_kernel), 160 * 1024, pu_flags, "ital_score_generic"))
                return rc;
    }
    if (hipEventRecord(ps->start, stream) != hipSuccess || hipStreamWaitEvent(ps->prep, ps->start, 0) != hipSuccess ||
        hipStreamWaitEvent(ps->main, ps->start, 0) != hipSuccess)
        return ital_fail(-5, "ital_score_generic: stream synchronisation failed");
    for (int64_t lo = 0; lo < d->n_cand; lo += S, ps->wide_slabs++) {
        const int buf = (int)(ps->wide_slabs & 1ull);
        double* base = d->work + (size_t)buf * (half + 1);
        g.slab_lo = lo;
        g.slab_n = d->n_cand - lo < S ? d->n_cand - lo : S;
        g.count = reinterpret_cast<unsigned int*>(base);
        g.meta = base + 1;
        g.recs = g.meta + g.slab_n * pl.total * 2;
        g.list = reinterpret_cast<unsigned int*>(g.recs + g.slab_n * pl.total * pl.R);
        g.cap = (unsigned int)(g.slab_n * pl.total);
        g.redo = pu ? reinterpret_cast<unsigned char*>(g.list + ((size_t)g.cap + 1) / 2 * 2) : nullptr;     // (slab_n bytes <= slab_n doubles)
        if (ps->summed_valid[buf]) (void)hipStreamWaitEvent(ps->prep, ps->summed[buf], 0);   // the buffer is free again (combined too)
        (void)hipMemsetAsync(g.count, 0, 2 * sizeof(unsigned int), ps->prep);
        if (pu) ITAL_LAUNCH(gen_prep_pu_kernel, dim3((unsigned)((g.slab_n + 1) / 2)), dim3(128), lds_pu, ps->prep, apu, g);
        ITAL_LAUNCH(gen_prep_kernel, dim3((unsigned)((g.slab_n * g.nsplit + 1) / 2)), dim3(128), lds_p, ps->prep, ap, g);
        (void)hipEventRecord(ps->built[buf], ps->prep);
        (void)hipStreamWaitEvent(ps->main, ps->built[buf], 0);
        ITAL_GEN_MAINS();
        ITAL_LAUNCH(gen_combine_kernel, dim3((unsigned)((g.slab_n + 3) / 4)), dim3(256), 0, ps->main, *d, g);
        (void)hipEventRecord(ps->summed[buf], ps->main);
        ps->summed_valid[buf] = true;
        const int rc = ital_check_launch("ital_score_generic(pipeline)");
        if (rc) return pipe_bail(ps, stream, rc);
    }
    (void)hipEventRecord(ps->combined, ps->main);
    if (d->defer_join) {
        ps->pending = true;       // joined by a later call of the step without the flag, or by ital_score_generic_join
        return 0;
    }
    (void)hipStreamWaitEvent(stream, ps->combined, 0);
    ps->pending = false;
    return 0;
#undef ITAL_GEN_MAINS
#undef ITAL_GEN_MAIN
}

extern "C" int ital_score_generic_join(hipStream_t stream) {
    PipeStreams* ps = pipe_streams();
    if (!ps) return ital_fail(-12, "ital_score_generic_join: cannot create the pipeline's streams");
    if (!ps->pending) return 0;
    (void)pipe_bail(ps, stream, 0);
    return ital_check_launch("ital_score_generic_join");
}
