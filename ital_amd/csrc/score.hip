// Mutual-information candidate scorer (reference ital/ital.py:183-224 `_call_iter_all`, :345-383 `prob_rel`,
// :432-450 `updated_prob_rel`; perfect-user feedback model ital.py:313-315).
//
//   MI(i) = sum_{r in {F,T}^t} w(r) * ( log(p'(r) + eps) - log(p(r) + eps) )
//     p (r) = P(sign pattern r) under N(mu_S, Sigma_S),  S = (batch so far, candidate i)   [natural order]
//     p'(r) = same after a simulated GP update with the labels r                         [sorted by data index]
//   label_estimation 'mean': w = p(r) and the terms are summed; 'optimistic': max term; 'pessimistic': min term.
//
// The simulated update is the closed form on the t x t block:
//     A = Sigma + noise*I,  W = A^-1,  G = Sigma W = I - noise*W,  mu' = mu + G (f - mu),  Sigma' = noise*G.
//
//   t = 1 : Phi closed form                     (one thread per candidate, HBM-bound)
//   t = 2 : Genz bivariate closed form          (one lane per (candidate, pattern))
//   t >= 3: Genz MVNDST randomised Korobov lattice as four kernels: generator state per candidate; one thread per
//           (candidate, sign pattern) prepares the call (COVSRT variable re-ordering, the 8 shifted lattices) into a record;
//           one wave per record sums the lattice, its points across the 64 lanes (FP64-VALU bound); one thread per
//           candidate adds the terms up.  The lattice shifts replay MVNUNI's stream at the offset the reference's *serial*
//           loop would reach for this (candidate, pattern, call): jump-ahead by 3x3 matrix powers, so the scores - and
//           the argmax - match the reference's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "device_math.h"
#include "ital_hip.h"
#include "ital_internal.h"
#include "qmc_common.h"
#include "qmc_exact.h"
#include "qmc_seed.h"
#include "select_common.h"

#ifndef ITAL_QMC_HOTK
#define ITAL_QMC_HOTK 1   // exp / log coefficients of the lattice loop as vector-register operands (device_math.h HotK)
#endif
#ifndef ITAL_QMC_MAIN_NH
// lattice items per lane and round of the lattice sum (x 2 chains: a point and its antithetic partner): three = six chains
// at every t.  More chains per lane fill the compacted tail branch of Phi^-1 better (15 % of the arguments: 58 of 64 lanes
// with six chains, 38 with four).  Until round 3 four chains were all the register file took at t = 4 (three waves per
// SIMD) and at t = 7, 8: the compiler kept the 39 coefficients of the tail branch in 78 vector registers for the whole
// kernel; with them materialised in place (device_math.h `lit_s`) the t = 4 kernel needs 126 registers for six chains.
// Measured (9298 x 256 / 25 000 x 256, `profiles/r3_chains_waves_variants.txt`): t = 4 2.10 -> 1.81 ms with six chains at
// four waves per SIMD (four chains at four waves 2.03, eight chains at three 2.03), t = 3 0.44 -> 0.42.
// t = 7, 8: four chains -- what fits THREE waves per SIMD (168 registers, per-stage coordinates, factor in LDS): occupancy
// is worth more there than fuller tail waves (25 000 x 512, `profiles/r3_k8_occupancy_variants.txt`: t = 8 steps 1216 ms
// with six chains at two waves, 1249 with four at two, 1068 with four at three; t = 7 341 / 304)
// Round 5: with the running product of the interval widths formed per stage (qmc_common.h ITAL_QMC_PIN_FF) the instantiations
// of t = 5 .. 8 lost 14 - 40 registers (145 / 167 / 162 / 161 -> 131 / 141 / 127 / 121) and the trade was measured again
// (25 000 x 512, k = 8, profiles/r5_variants_wide2.txt, ms per step): t = 6 four chains at FOUR waves 71.2 (six at three: 73.7),
// t = 7 six chains at three waves 280.8 (four at three: 303.0, four at four: 284.2), t = 8 four chains at four waves 965
// (four at three: 1022, six at three: 977); t = 5 indifferent (21.2 - 21.8)
#define ITAL_QMC_MAIN_NH(T) ((T) == 6 || (T) == 8 ? 2 : 3)
#endif
#ifndef ITAL_QMC_WAVES
// waves per SIMD the register allocation aims at: four up to t = 4 (126 registers), three at t = 5, 6 (145 / 167: per step
// 24.8 -> 21.1 ms, 95 -> 80 ms at 25 000 candidates) and at t = 7, 8 (168 each with four chains).  (Five at t = 3 fits
// without scratch only without the width emulation of negated variables -- flip_width, qmc_common.h -- and was worth
// 2 %: 0.386 -> 0.377 ms; with it 12 B of scratch: four.)
// Round 5 (see ITAL_QMC_MAIN_NH): four also at t = 6 and t = 8 (110 / 121 registers with four chains).
#define ITAL_QMC_WAVES(T) ((T) <= 4 || (T) == 6 || (T) == 8 ? 4 : 3)
#endif
#ifndef ITAL_QMC_MAIN_PS
#define ITAL_QMC_MAIN_PS(T) ((T) >= 7)   // lattice coordinates formed per stage (qmc_lane_sum_ps): T = 7, 8 spill otherwise
#endif
#ifndef ITAL_QMC_MAIN_KE6
#define ITAL_QMC_MAIN_KE6(T) ((T) == 8)  // six of the ten exp coefficients in registers, four in place: t = 8 at three waves, no scratch
#endif
#ifndef ITAL_QMC_MAIN_KEN
#define ITAL_QMC_MAIN_KEN(T) ((T) == 7 ? 2 : -1)   // >= 0: that many of the exp coefficients in registers (HotKEn), the others
                                                  // in place: t = 7 spills 21 registers with all ten, 12 with six, none with two
#endif
#ifndef ITAL_QMC_MAIN_KE
#define ITAL_QMC_MAIN_KE(T) 1            // only the exp coefficients as register operands: the logarithm of the tail branch
                                         // takes its own in place (lit_s), HotK's nine log coefficients would sit in 18
                                         // vector registers unused
#endif
#ifndef ITAL_QMC_MAIN_KS
#define ITAL_QMC_MAIN_KS(T) 0            // exp / log coefficients as scalar-register operands (measured: +7 % at t = 8, dropped)
#endif
#ifndef ITAL_QMC_MAIN_CFL
#define ITAL_QMC_MAIN_CFL(T) ((T) >= 7)  // ... with the factor read from LDS at use instead of held in scalar registers
#endif

namespace ital {

struct ScoreArgs {
    int t;
    int64_t n_cand;
    const int32_t* cand;    // local row per list position
    const uint8_t* alive;   // position still a candidate
    const double* mu;       // [n_rows]
    const double* s2;       // [n_rows] unclamped posterior variance
    const double* C;        // [t-1][ldc] cross-covariance columns of the batch members
    int64_t ldc;
    int64_t row_offset;     // global data index of local row 0
    int64_t pos_offset;     // global list position of local position 0 (when gpos == nullptr)
    const int64_t* gpos;    // [n_cand] global list position of every local position, or nullptr
    ital_batch b;
    double noise, eps;
    int label_mode;         // 0 mean, 1 optimistic, 2 pessimistic
    double* mi;             // [n_cand]
    // MVNUNI replay
    int seed[6];            // generator state at the first call of this greedy step
    const long long* jump;  // [48][18]: transition matrices for 2^b calls (of this step's dimension)
    const long long* jumppat;   // [2^t][18]: transition matrices for 2r calls (the prior call of sign pattern r)
    const double* vk;       // [t-1] Korobov generators
    int* status;
    SelectTail sel;         // optional: the step's selection as the tail of its last scoring launch (sel.enabled)
};

__device__ __forceinline__ double log_eps(double p, double eps) { return log(p + eps); }

__device__ __forceinline__ void mi_accumulate(double& mi, double pr, double pu, double eps, int mode) {
    double cur = log_eps(pu, eps) - log_eps(pr, eps);  // perfect user: likelihood weight 1 (ital.py:208, fb_mc_num = 1)
    if (mode == 1) {
        if (cur > mi) mi = cur;
    } else if (mode == 2) {
        if (mi == 0 || cur < mi) mi = cur;
    } else {
        mi += cur * pr;
    }
}

// ------------------------------------------------------------------------------------------------ t = 1
__global__ __launch_bounds__(256) void score_t1_kernel(ScoreArgs a) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < a.n_cand && a.alive[p];
    double mi = 0.0;
    if (valid) {
        const int row = a.cand[p];
        const double mu = a.mu[row];
        const double su = a.s2[row];               // unclamped: what the simulated update sees (gp.py:334)
        const double sc = fmax(0.0, su);           // clamped: predict_stored(cov_mode='diag') (gp.py:229, ital.py:558)
        const double p_irr = norm_cdf0(mu, sqrt(sc));
        const double w = 1.0 / (su + a.noise);
        const double g = su * w;
        const double s_upd = a.noise * g;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const double f = r ? 1.0 : -1.0;
            const double mu_upd = mu + g * (f - mu);
            const double q = norm_cdf0(mu_upd, sqrt(s_upd));
            const double pr = r ? 1.0 - p_irr : p_irr;
            const double pu = r ? 1.0 - q : q;
            mi_accumulate(mi, pr, pu, a.eps, a.label_mode);
        }
        a.mi[p] = mi;
    }
    if (a.sel.enabled) {
        Best c = {mi, valid ? (a.gpos ? a.gpos[p] : a.pos_offset + p) : -1, p};
        c = block_best(c, 0);
        select_tail(a.sel, c, blockIdx.x, gridDim.x, true, gridDim.x);
    }
}

// ------------------------------------------------------------------------------------------------ t = 2
// A block scores 32 candidates: 128 (candidate, pattern) pairs, pattern r = 0..3 in itertools.product order (bit 1 = batch
// member, bit 0 = candidate).  Waves 0-1 evaluate the pairs' prior probabilities, waves 2-3 the probabilities after the
// simulated update -- one bivariate closed form per thread, the same branch for a whole wave (the step is a few hundred waves
// of one long dependent chain each: this way twice as many, half as long); 32 threads then add the terms up in order.
__global__ __launch_bounds__(256) void score_t2_kernel(ScoreArgs a) {
    __shared__ double vals[2][128];
    const int pair = threadIdx.x & 127;
    const bool updated = threadIdx.x >= 128;
    const int64_t p = (int64_t)blockIdx.x * 32 + (pair >> 2);
    const int r = pair & 3;
    const bool valid = p < a.n_cand && a.alive[p];
    double val = updated ? 1.0 : 0.0;
    if (valid) {
        const int row = a.cand[p];
        const double m0 = a.b.bmu[0], m1 = a.mu[row];
        const double s00 = a.b.sig[0], s11 = a.s2[row], s01 = a.C[row];
        const bool rel0 = (r >> 1) & 1, rel1 = r & 1;
        if (!updated) {
            const double sd0 = sqrt(s00), sd1 = sqrt(s11);
            val = bvn_orthant(-m0 / sd0, -m1 / sd1, rel0, rel1, s01 / (sd0 * sd1));
        } else {
            // simulated update, closed form
            const double a00 = s00 + a.noise, a11 = s11 + a.noise, a01 = s01;
            const double det = a00 * a11 - a01 * a01;
            const double w00 = a11 / det, w11 = a00 / det, w01 = -a01 / det;
            const double g00 = 1.0 - a.noise * w00, g11 = 1.0 - a.noise * w11, g01 = -a.noise * w01;
            const double f0 = rel0 ? 1.0 : -1.0, f1 = rel1 ? 1.0 : -1.0;
            const double u0 = m0 + g00 * (f0 - m0) + g01 * (f1 - m1);
            const double u1 = m1 + g01 * (f0 - m0) + g11 * (f1 - m1);
            const double c00 = a.noise * g00, c11 = a.noise * g11, c01 = a.noise * g01;
            const double t0 = sqrt(c00), t1 = sqrt(c11);
            double h = -u0 / t0, k = -u1 / t1;
            const double rho = c01 / (t0 * t1);
            // variables sorted by data index (ital.py:448): symmetric for the closed form, kept for the bookkeeping
            const int64_t gi = a.row_offset + row;
            if (gi < a.b.bidx[0]) val = bvn_orthant(k, h, rel1, rel0, rho);
            else val = bvn_orthant(h, k, rel0, rel1, rho);
        }
    }
    vals[updated ? 1 : 0][pair] = val;
    __syncthreads();
    Best c = {0.0, -1, 0};
    if (threadIdx.x < 32) {
        const int64_t pc = (int64_t)blockIdx.x * 32 + threadIdx.x;
        if (pc < a.n_cand && a.alive[pc]) {
            double mi = 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) mi_accumulate(mi, vals[0][4 * threadIdx.x + q], vals[1][4 * threadIdx.x + q], a.eps, a.label_mode);
            a.mi[pc] = mi;
            c.val = mi;
            c.pos = a.gpos ? a.gpos[pc] : a.pos_offset + pc;
            c.loc = pc;
        }
    }
    if (a.sel.enabled) {
        c = block_best(c, 0);
        select_tail(a.sel, c, blockIdx.x, gridDim.x, true, gridDim.x);
    }
}

// ------------------------------------------------------------------------------------------------ t >= 3
// Four kernels per greedy step, every one with all 64 lanes of its waves at work:
//   qmc_seed_kernel    thread per candidate         MVNUNI state at the candidate's first call (jump-ahead)
//   qmc_prep_kernel<T> thread per (candidate, r)    candidate-level algebra, the standardised prior problem of sign pattern r,
//                                                   COVSRT, saturation verdicts, the call's 8 shifted lattices -> one record
//   qmc_main_kernel<T> wave per (candidate, r)      the lattice sum of the prior call (the FP64-VALU bound part), MI term of r
//   qmc_combine_kernel thread per candidate         terms of the 2^T patterns in itertools.product order -> mi
// The records travel through a workspace in HBM (208 B per call at T = 4); the candidates are processed in slabs that fit it.
template <int T>
struct Qmc {
    static constexpr int NDIM = T - 1;
    static constexpr int NP = NDIM < 10 ? NDIM : 10;
    static constexpr int PRIME = P_TAB[NP - 1];
    static constexpr int NCOV = T * (T + 1) / 2;
    static constexpr int NCOR = T * (T - 1) / 2;
    static constexpr int NPAT = 1 << T;                       // sign patterns = prior calls per candidate
    static constexpr int NCALLS = 2 << T;                     // calls the reference makes per candidate (prior + updated)
    static constexpr int LAT = 8 * NDIM * 2;                  // per call: permuted generators + shifts, 8 shifts (in LDS)
    // in the record the 8 NDIM shifts travel as MVNUNI's 32-bit integers and the 8 NDIM generators as byte indices into
    // the generator vector (the lattice-sum kernel forms the identical doubles): 5 NDIM doubles instead of 16 NDIM
    static constexpr int LATP = 5 * NDIM;
    static constexpr int R_LIM = NCOR, R_META = NCOR + T, R_LAT = NCOR + T + 1;
    static constexpr int REC = R_LAT + LATP;                  // doubles per prepared call
    static constexpr int SLAB_RAW = NCOV + 2 * T + NDIM;      // prep scratch per thread: packed factor, limits, expected values, generators
    static constexpr int SLAB = SLAB_RAW | 1;                 // odd stride: conflict-free per-thread slabs
    static constexpr int PREP_THREADS = T <= 6 ? 256 : 128;
    // lattice items per lane and round of the lattice-sum kernel (each with its antithetic partner), see ITAL_QMC_MAIN_NH
    static constexpr int NH = ITAL_QMC_MAIN_NH(T);
    static constexpr int TAILQ = 128 * NH;                    // compaction queue of the Phi^-1 tail branch (in place)
    static constexpr bool PS = ITAL_QMC_MAIN_PS(T), CFL = PS && ITAL_QMC_MAIN_CFL(T);
    static constexpr int WAVE_DOUBLES = LAT + TAILQ + (CFL ? NCOR + T : 0);
    // exp / log coefficients: vector-register operands, or scalar ones where the factor does not occupy the scalar file
    typedef typename std::conditional<(CFL && ITAL_QMC_MAIN_KS(T)), HotKS,
                typename std::conditional<(ITAL_QMC_MAIN_KEN(T) >= 0), HotKEn<(ITAL_QMC_MAIN_KEN(T) >= 0 ? ITAL_QMC_MAIN_KEN(T) : 0)>,
                typename std::conditional<(ITAL_QMC_MAIN_KE6(T)), HotKE6,
                    typename std::conditional<(ITAL_QMC_MAIN_KE(T)), HotKE, HotK>::type>::type>::type>::type Coef;
    static constexpr int64_t CAND_DOUBLES = (int64_t)NPAT * (REC + 1) + 3;   // records, terms, 6 ints of generator state
};
// meta word of a record: bit 0 evaluate the lattice sum; bit 1 prior probability == 1 (else 0) when not evaluated;
// bit 2 probability after the simulated update == 1 (else 0); bits 8.. limit types after COVSRT
constexpr long long META_EVAL = 1, META_PRIOR_ONE = 2, META_POST_ONE = 4;
constexpr long long META_EXACT = 8;   // set by the lattice sum: recompute in MVKBRV's own summation order (qmc_exact.h)

// Swap rows/columns p < q of the packed lower-triangular matrix, the limits and the limit-type bits (RCSWP).
template <int T>
__device__ void rcswp(int p, int q, double* cov, double* lim, unsigned& infi) {
    double tmp = lim[p]; lim[p] = lim[q]; lim[q] = tmp;
    unsigned bp = (infi >> p) & 1u, bq = (infi >> q) & 1u;
    infi = (infi & ~((1u << p) | (1u << q))) | (bq << p) | (bp << q);
    tmp = cov[pidx(p, p)]; cov[pidx(p, p)] = cov[pidx(q, q)]; cov[pidx(q, q)] = tmp;
    for (int j = 0; j < p; j++) { tmp = cov[pidx(p, j)]; cov[pidx(p, j)] = cov[pidx(q, j)]; cov[pidx(q, j)] = tmp; }
    for (int i = p + 1; i < q; i++) { tmp = cov[pidx(i, p)]; cov[pidx(i, p)] = cov[pidx(q, i)]; cov[pidx(q, i)] = tmp; }
    for (int i = q + 1; i < T; i++) { tmp = cov[pidx(i, p)]; cov[pidx(i, p)] = cov[pidx(i, q)]; cov[pidx(i, q)] = tmp; }
}

// COVSRT: order the variables by increasing expected interval probability, build the (row-scaled) Cholesky factor.
// Limit types are one-sided here (bit = 1: [lim, inf), bit = 0: (-inf, lim]).  Returns false on a singular factor.
template <int T>
__device__ bool covsrt(double* cov, double* lim, double* y, unsigned& infi) {
    const double SQTWPI = 2.506628274631001, EPS = 1e-10;
    bool ok = true;
    for (int i = 0; i < T; i++) {
        double dmin = 0, emin = 1, zmin = 0, cvdiag = 0;
        int jmin = i;
        for (int j = i; j < T; j++) {
            const double cjj = cov[pidx(j, j)];
            if (cjj > EPS) {
                const double sumsq = sqrt_pos(cjj);
                double sum = 0;
                for (int k = 0; k < i; k++) sum += cov[pidx(j, k)] * y[k];
                const double z = fast_div(lim[j] - sum, sumsq);
                const double ph = mvn_phi(z);
                const bool lower = (infi >> j) & 1u;
                const double d = lower ? ph : 0.0;
                const double e = lower ? 1.0 : fmax(ph, 0.0);
                if (emin + d >= e + dmin) { jmin = j; zmin = z; dmin = d; emin = e; cvdiag = sumsq; }
            }
        }
        if (jmin > i) rcswp<T>(i, jmin, cov, lim, infi);
        cov[pidx(i, i)] = cvdiag;
        if (cvdiag > 0) {
            const double rdiag = fast_div(1.0, cvdiag);   // cvdiag in (1e-5, 1]: well scaled
            for (int l = i + 1; l < T; l++) {
                cov[pidx(l, i)] = cov[pidx(l, i)] * rdiag;
                for (int j = i + 1; j <= l; j++) cov[pidx(l, j)] -= cov[pidx(l, i)] * cov[pidx(j, i)];
            }
            const bool lower = (infi >> i) & 1u;
            if (emin > dmin + EPS) {
                const double dens = -exp_neg(-zmin * zmin / 2) * (1.0 / SQTWPI);
                const double yl = lower ? dens : 0.0, yu = lower ? 0.0 : dens;
                y[i] = fast_div(yu - yl, emin - dmin);
            } else {
                y[i] = zmin;
            }
            for (int j = 0; j <= i; j++) cov[pidx(i, j)] = cov[pidx(i, j)] * rdiag;
            lim[i] = lim[i] * rdiag;
        } else {
            ok = false;
            y[i] = 0;
        }
    }
    return ok;
}

// Verdict on an orthant problem from its standardised limits alone (no COVSRT): whatever order COVSRT picks, the
// conditional limit of variable a is (lim_a - sum_j c_aj y_j) / c_aa with sum_j c_aj^2 + c_aa^2 = 1 and |y_j| <= 9, so it
// stays beyond +-37 (where MVNPHI is exactly 0 or 1) once |lim_a| > 37 + 9 sqrt(T - 1).  l[a]: the limit in the variable's
// own direction (interval [l, inf)).  Returns 2 (probability exactly 1), 4 (exactly 0) or 0 (undecided).
template <int T>
__device__ __forceinline__ int early_verdict(const double (&l)[T]) {
    const double thr = 37.0 + 9.0 * sqrt((double)(T - 1));
    bool all_full = true, any_empty = false;
#pragma unroll
    for (int a = 0; a < T; a++) {
        if (l[a] > thr) any_empty = true;
        if (!(l[a] < -thr)) all_full = false;
    }
    return any_empty ? 4 : (all_full ? 2 : 0);
}

__device__ __forceinline__ int64_t list_position(const ScoreArgs& a, int64_t p) { return a.gpos ? a.gpos[p] : a.pos_offset + p; }

// MVNUNI state at the first call of every candidate of the slab: the step's seed advanced by
// (rank of the candidate among the live list positions) * ncalls calls -- one 3x3 matrix product mod m per set bit.
__global__ __launch_bounds__(256) void qmc_seed_kernel(SeedArgs s) {
    qmc_seed_body(s, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

static SeedArgs seed_args(const ScoreArgs& a, int ncalls, int64_t slab_lo, int64_t slab_n, int* seeds) {
    SeedArgs s = {a.alive, a.gpos, a.pos_offset, a.b.bgpos, a.t - 1, {a.seed[0], a.seed[1], a.seed[2], a.seed[3], a.seed[4], a.seed[5]},
                  a.jump, ncalls, slab_lo, slab_n, seeds};
    return s;
}

template <int T>
__global__ __launch_bounds__(Qmc<T>::PREP_THREADS) void qmc_prep_kernel(ScoreArgs a, int64_t slab_lo, int64_t slab_n,
                                                                         const int* __restrict__ seeds,
                                                                         double* __restrict__ recs) {
    using Q = Qmc<T>;
    extern __shared__ double lds_all[];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slab_n) return;
    const int64_t p = slab_lo + i;
    if (!a.alive[p]) return;
    const int r = blockIdx.y;                   // sign pattern: bit T-1-v = variable v relevant (itertools.product order)
    const int row = a.cand[p];
    double* rec = recs + ((int64_t)i * Q::NPAT + r) * Q::REC;

    // ---------------- candidate-level quantities
    double mean[T], Sg[T][T];
#pragma unroll
    for (int v = 0; v < T - 1; v++) {
        mean[v] = a.b.bmu[v];
#pragma unroll
        for (int j = 0; j < T - 1; j++) Sg[v][j] = a.b.sig[v * a.b.kmax + j];
        const double c = a.C[(int64_t)v * a.ldc + row];
        Sg[v][T - 1] = c;
        Sg[T - 1][v] = c;
    }
    mean[T - 1] = a.mu[row];
    Sg[T - 1][T - 1] = a.s2[row];  // not clamped (gp.py:254)
    double sd[T], own[T];
    long long meta = 0;
    // ---------------- the probability after the simulated update with the labels r: closed form on the t x t block,
    // W = (Sigma + noise I)^-1 through its Cholesky factor, G = I - noise W, mu' = mu + G (f - mu), Sigma' = noise G.
    // With the perfect user every variable lands ~1/sqrt(noise) standard deviations on its own side: decided here.
    {
        double Lc[T][T];
#pragma unroll
        for (int v = 0; v < T; v++)
#pragma unroll
            for (int j = 0; j <= v; j++) {
                double x = Sg[v][j] + (v == j ? a.noise : 0.0);
#pragma unroll
                for (int q = 0; q < j; q++) x -= Lc[v][q] * Lc[j][q];
                Lc[v][j] = (v == j) ? sqrt(x) : x / Lc[j][j];
            }
        double Li[T][T];  // inverse of the factor (lower)
#pragma unroll
        for (int j = 0; j < T; j++) {
#pragma unroll
            for (int v = 0; v < T; v++) {
                if (v < j) { Li[v][j] = 0; continue; }
                double x = (v == j) ? 1.0 : 0.0;
#pragma unroll
                for (int q = j; q < v; q++) x -= Lc[v][q] * Li[q][j];
                Li[v][j] = x / Lc[v][v];
            }
        }
#pragma unroll
        for (int v = 0; v < T; v++) {
            double mu_u = mean[v], gvv = 0;
#pragma unroll
            for (int j = 0; j < T; j++) {
                double w = 0;
#pragma unroll
                for (int q = (v > j ? v : j); q < T; q++) w += Li[q][v] * Li[q][j];
                const double gvj = (v == j ? 1.0 : 0.0) - a.noise * w;
                const double f = ((r >> (T - 1 - j)) & 1) ? 1.0 : -1.0;
                mu_u += gvj * (f - mean[j]);
                if (j == v) gvv = gvj;
            }
            const double lim_u = -mu_u / sqrt(a.noise * gvv);
            own[v] = ((r >> (T - 1 - v)) & 1) ? lim_u : -lim_u;
        }
        const int e = early_verdict<T>(own);
        if (e == 2) meta |= META_POST_ONE;
        else if (e == 0) atomicOr(a.status, 4);   // an updated probability that needs the integrator: the general scorer's job
    }
    // ---------------- the prior probability of pattern r: standardised problem in natural order (ital.py:373-383)
    double* cov = lds_all + (size_t)threadIdx.x * Q::SLAB;
    double* lim = cov + Q::NCOV;
    double* y = lim + T;
    double* gen = y + T;
    unsigned infi = 0;
#pragma unroll
    for (int v = 0; v < T; v++) {
        sd[v] = sqrt(Sg[v][v]);
        const double piv = -mean[v] / sd[v];
        const unsigned bit = (r >> (T - 1 - v)) & 1;
        lim[v] = piv;
        own[v] = bit ? piv : -piv;
        infi |= bit << v;
    }
    int verdict = early_verdict<T>(own);
    bool evaluate = false;
    if (verdict == 0) {
#pragma unroll
        for (int v = 0; v < T; v++) {
#pragma unroll
            for (int j = 0; j < v; j++) cov[pidx(v, j)] = Sg[v][j] / (sd[v] * sd[j]);
            cov[pidx(v, v)] = 1.0;
        }
        const bool okc = covsrt<T>(cov, lim, y, infi);
        if (!okc) {
            atomicOr(a.status, 2);  // singular conditional covariance: not supported by this kernel (general scorer)
        } else {
            // integrand identically 1 / 0?  every conditional limit stays beyond +-37 for any |y| <= 9
            bool sat1 = true, sat0 = false;
            for (int v = 0; v < T; v++) {
                double bound = 0;
                for (int j = 0; j < v; j++) bound += fabs(cov[pidx(v, j)]) * 9.0;
                const bool lower = (infi >> v) & 1u;
                if (lower) {
                    if (!(lim[v] + bound < -37.0)) sat1 = false;
                    if (lim[v] - bound > 37.0) sat0 = true;
                } else {
                    if (!(lim[v] - bound > 37.0)) sat1 = false;
                    if (lim[v] + bound < -37.0) sat0 = true;
                }
            }
            if (sat0) verdict = 4;
            else if (sat1) verdict = 2;
            else evaluate = true;
        }
    }
    if (verdict == 2) meta |= META_PRIOR_ONE;
    if (evaluate) {
        meta |= META_EVAL | ((long long)infi << 8);
        // (ITAL_QMC_FLIP: the record holds the call with every variable bounded above -- a variable bounded below enters
        // negated: its limit, its row and its column of the factor change sign; the lattice sum moves its shifts by 1/2)
        const unsigned fl = ITAL_QMC_FLIP ? infi : 0u;
        for (int v = 1; v < T; v++)
            for (int j = 0; j < v; j++)
                rec[v * (v - 1) / 2 + j] = (((fl >> v) ^ (fl >> j)) & 1u) ? -cov[pidx(v, j)] : cov[pidx(v, j)];
        for (int v = 0; v < T; v++) rec[Q::R_LIM + v] = ((fl >> v) & 1u) ? -lim[v] : lim[v];
        // The call's 8 randomly shifted lattices.  A call draws 8*(2*NDIM-1) uniforms from MVNUNI whether it is evaluated or
        // not: per shift NDIM-1 for DKSMRC's random transposition of the generator vector (the transpositions accumulate
        // from shift to shift), then NDIM shifts.  The prior call of pattern r is call 2r of the candidate.
        const int* sp = seeds + i * 6;
        MrgState sti = {sp[0], sp[1], sp[2], sp[3], sp[4], sp[5]};
        mrg_apply(sti, a.jumppat + (int64_t)r * 18);   // row r: 2r calls
        MrgStateF st = mrg_to_f(sti);
        for (int j = 0; j < Q::NDIM; j++) gen[j] = (double)j;                  // positions in the generator vector a.vk
        unsigned int* shifts = reinterpret_cast<unsigned int*>(rec + Q::R_LAT);
        unsigned char* perm = reinterpret_cast<unsigned char*>(rec + Q::R_LAT + 4 * Q::NDIM);
        for (int sft = 0; sft < 8; sft++) {
            for (int j = 1; j <= Q::NDIM - 1; j++) {
                const double u = mrg_next_f(st);
                const int jp = (int)(j + u * (Q::NDIM + 1 - j));
                const double xt = gen[j - 1];
                gen[j - 1] = gen[jp - 1];
                gen[jp - 1] = xt;
            }
            for (int j = 0; j < Q::NDIM; j++) perm[sft * Q::NDIM + j] = (unsigned char)gen[j];
            for (int j = 0; j < Q::NDIM; j++) shifts[sft * Q::NDIM + j] = (unsigned int)mrg_next_z(st);
        }
    }
    rec[Q::R_META] = __longlong_as_double(meta);
}

template <int T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ITAL_QMC_WAVES(T), ITAL_QMC_WAVES(T)))) void qmc_main_kernel(
    const uint8_t* __restrict__ alive, int64_t slab_lo, int64_t slab_n, const double* __restrict__ recs,
    const double* __restrict__ vk, double eps, int label_mode, double* __restrict__ terms) {
    using Q = Qmc<T>;
    extern __shared__ double lds_all[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t item = (int64_t)blockIdx.x * 4 + wid;          // (candidate of the slab, sign pattern)
    const int64_t i = item >> T;
    if (i >= slab_n) return;
    if (!alive[slab_lo + i]) return;
    const double* rec = recs + item * Q::REC;
    const long long meta = __double_as_longlong(uniform_f64(rec[Q::R_META]));
    double pr = (meta & META_PRIOR_ONE) ? 1.0 : 0.0;
    if (meta & META_EVAL) {
        double* lat = lds_all + (size_t)wid * Q::WAVE_DOUBLES;
        double* tailq = lat + Q::LAT;
        const unsigned flips = ITAL_QMC_FLIP ? (unsigned)(meta >> 8) & ((1u << T) - 1u) : 0u;   // variables bounded below
        {   // unpack the lattices: generator = vk[index], shift = integer * 1/(m1 + 1) exactly as MVNUNI forms it
            const unsigned int* shifts = reinterpret_cast<const unsigned int*>(rec + Q::R_LAT);
            const unsigned char* perm = reinterpret_cast<const unsigned char*>(rec + Q::R_LAT + 4 * Q::NDIM);
            for (int q = lane; q < 8 * Q::NDIM; q += 64) {
                lat[q] = vk[perm[q]];
                lat[8 * Q::NDIM + q] = (double)shifts[q] * MRG_INVMP1 + (((flips >> (q % Q::NDIM)) & 1u) ? 0.5 : 0.0);
            }
        }
        double cf[Q::NCOR > 0 ? Q::NCOR : 1], lm[T];
        double* slab = tailq + Q::TAILQ;
        if (Q::CFL) {
            for (int q = lane; q < Q::NCOR + T; q += 64) slab[q] = rec[q];     // factor, then limits (R_LIM == NCOR)
        } else {
#pragma unroll
            for (int q = 0; q < Q::NCOR; q++) cf[q] = uniform_f64(rec[q]);
#pragma unroll
            for (int q = 0; q < T; q++) lm[q] = uniform_f64(rec[Q::R_LIM + q]);
        }
        const unsigned infi = (unsigned)(meta >> 8) & ((1u << T) - 1u);   // FLIP: the variables that entered negated
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if ITAL_QMC_HOTK
        typename Q::Coef kk;
        kk.load();
        constexpr bool FL = ITAL_QMC_FLIP != 0;
        const double acc = Q::PS ? qmc_lane_sum_ps<T, typename Q::Coef, Q::NH, Q::CFL, FL>(lat, cf, lm, slab, infi, tailq, lane, kk)
                                 : qmc_lane_sum<T, typename Q::Coef, Q::NH, FL>(lat, cf, lm, infi, tailq, lane, kk);
#else
        const double acc = qmc_lane_sum<T, LitK, Q::NH, ITAL_QMC_FLIP != 0>(lat, cf, lm, infi, tailq, lane);
#endif
        pr = wave_sum(acc) / (16.0 * Q::PRIME);
        // label_estimation 'optimistic' / 'pessimistic' compare terms for exact equality: a sum this close to 0 or 1 is
        // formed again in the reference's own order (qmc_exact_kernel, right after this launch)
#ifndef ITAL_NO_EXACT_FLAG      // (A/B switch: the flag costs nothing measurable, profiles/r4_exact_flag_ab.txt)
        if (label_mode != 0 && (pr > 1.0 - EXACT_BAND || pr < EXACT_BAND) && lane == 0)
            const_cast<double*>(rec)[Q::R_META] = __longlong_as_double(meta | META_EXACT);
#endif
    }
    if (lane == 0) {
        const double pu = (meta & META_POST_ONE) ? 1.0 : 0.0;
        const double cur = log_eps(pu, eps) - log_eps(pr, eps);   // perfect user: likelihood weight 1 (ital.py:208)
        terms[item] = label_mode == 0 ? cur * pr : cur;
    }
}

// The flagged calls of a slab again, in the reference's summation order (qmc_exact.h): wave per (candidate, pattern); all
// but a handful leave at once.  Launched only with label_estimation 'optimistic' / 'pessimistic'.
template <int T>
__global__ __launch_bounds__(64) void qmc_exact_kernel(const uint8_t* __restrict__ alive, int64_t slab_lo, int64_t slab_n,
                                                       const double* __restrict__ recs, const double* __restrict__ vk, double eps,
                                                       int label_mode, double* __restrict__ terms) {
    using Q = Qmc<T>;
    extern __shared__ double lds_all[];
    const int lane = threadIdx.x;
    const int64_t item = blockIdx.x;
    const int64_t i = item >> T;
    if (i >= slab_n || !alive[slab_lo + i]) return;
    const double* rec = recs + item * Q::REC;
    const long long meta = __double_as_longlong(uniform_f64(rec[Q::R_META]));
    if (!(meta & META_EXACT)) return;
    double* slab = lds_all;                       // packed factor with (unused) diagonal, limits: the natural form
    double* lat = slab + Q::NCOV + T;
    double* tailq = lat + Q::LAT;
    double* vals = tailq + 128;
    const unsigned infi = (unsigned)(meta >> 8) & ((1u << T) - 1u);
    const unsigned fl = ITAL_QMC_FLIP ? infi : 0u;       // the record holds the variables bounded below negated: undo
    for (int q = lane; q < Q::NCOV; q += 64) {
        int row = 0;
        while ((row + 1) * (row + 2) / 2 <= q) row++;
        const int col = q - row * (row + 1) / 2;
        double v = 0.0;
        if (col < row) {
            v = rec[row * (row - 1) / 2 + col];
            if (((fl >> row) ^ (fl >> col)) & 1u) v = -v;
        }
        slab[q] = v;
    }
    for (int q = lane; q < T; q += 64) slab[Q::NCOV + q] = ((fl >> q) & 1u) ? -rec[Q::R_LIM + q] : rec[Q::R_LIM + q];
    {
        const unsigned int* shifts = reinterpret_cast<const unsigned int*>(rec + Q::R_LAT);
        const unsigned char* perm = reinterpret_cast<const unsigned char*>(rec + Q::R_LAT + 4 * Q::NDIM);
        for (int q = lane; q < 8 * Q::NDIM; q += 64) {
            lat[q] = vk[perm[q]];
            lat[8 * Q::NDIM + q] = (double)shifts[q] * MRG_INVMP1;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    qmc_point_values<T>(T, slab, infi, (1u << T) - 1u, lat, lane, tailq, vals);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double pr = mvkbrv_serial(Q::PRIME, vals, lane);
    if (lane == 0) {
        const double pu = (meta & META_POST_ONE) ? 1.0 : 0.0;
        const double cur = log_eps(pu, eps) - log_eps(pr, eps);
        terms[item] = label_mode == 0 ? cur * pr : cur;
    }
}

// mi[p] from the terms of the 2^T patterns, in pattern order (the reference's loop, ital.py:207-222); with `sel` the
// blocks also leave their best candidate behind and the step's last block selects (select_tail).
__global__ __launch_bounds__(256) void qmc_combine_kernel(const double* __restrict__ terms, const uint8_t* __restrict__ alive,
                                                          int64_t slab_lo, int64_t slab_n, int npat, int label_mode,
                                                          double* __restrict__ mi, int64_t pos_offset,
                                                          const int64_t* __restrict__ gpos, SelectTail sel, int part0,
                                                          int nparts, int finishing) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < slab_n && alive[slab_lo + i];
    double s = 0.0;
    if (valid) {
        for (int r = 0; r < npat; r++) {
            const double cur = terms[i * npat + r];
            if (label_mode == 1) { if (cur > s) s = cur; }
            else if (label_mode == 2) { if (s == 0 || cur < s) s = cur; }
            else s += cur;
        }
        mi[slab_lo + i] = s;
    }
    if (sel.enabled) {
        const int64_t p = slab_lo + i;
        Best c = {s, valid ? (gpos ? gpos[p] : pos_offset + p) : -1, p};
        c = block_best(c, 0);
        select_tail(sel, c, part0 + blockIdx.x, nparts, finishing != 0, gridDim.x);
    }
}

template <int T>
static int launch_qmc(const ScoreArgs& a, double* work, int64_t work_doubles, int64_t sel_parts_len, hipEvent_t ev0, hipEvent_t ev1,
                      hipStream_t stream, bool seeds_ready, SeedArgs* seeds_only) {
    using Q = Qmc<T>;
    int64_t slab = work_doubles / Q::CAND_DOUBLES;
    if (slab < 1) return ital_fail(-12, "ital_score_step: workspace smaller than one candidate (see ital_score_workspace)");
    if (slab > a.n_cand) slab = a.n_cand;
    const size_t lds_main = (size_t)4 * Q::WAVE_DOUBLES * sizeof(double);
    const size_t lds_prep = (size_t)Q::PREP_THREADS * Q::SLAB * sizeof(double);
    static ItalLdsFlags lds_flags;
    if (lds_prep > 48 * 1024) {
        const int rc = ital_raise_lds_limit(reinterpret_cast<const void*>(&qmc_prep_kernel<T>), (int)lds_prep, lds_flags,
                                            "ital_score_step");
        if (rc) return rc;
    }
    double* recs = work;
    double* terms = recs + slab * Q::NPAT * Q::REC;
    int* seeds = reinterpret_cast<int*>(terms + slab * Q::NPAT);
    if (seeds_only) {
        // only tell where the seeds of this step go (one slab: the caller computes them in another launch)
        if (slab < a.n_cand) return 1;
        *seeds_only = seed_args(a, Q::NCALLS, 0, a.n_cand, seeds);
        return 0;
    }
    int nparts = 0, part0 = 0;
    for (int64_t lo = 0; lo < a.n_cand; lo += slab) nparts += (int)(((a.n_cand - lo < slab ? a.n_cand - lo : slab) + 255) / 256);
    if (a.sel.enabled && 3 * (int64_t)nparts > sel_parts_len)
        return ital_fail(-22, "ital_score_step: sel_parts too small for the blocks of this step");
    for (int64_t lo = 0; lo < a.n_cand; lo += slab) {
        const int64_t n = a.n_cand - lo < slab ? a.n_cand - lo : slab;
        if (!(seeds_ready && slab >= a.n_cand))
            ITAL_LAUNCH(qmc_seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, seed_args(a, Q::NCALLS, lo, n, seeds));
        ITAL_LAUNCH(qmc_prep_kernel<T>, dim3((unsigned)((n + Q::PREP_THREADS - 1) / Q::PREP_THREADS), Q::NPAT),
                           dim3(Q::PREP_THREADS), lds_prep, stream, a, lo, n, seeds, recs);
        if (ev0 && lo == 0) (void)hipEventRecord(ev0, stream);
        ITAL_LAUNCH(qmc_main_kernel<T>, dim3((unsigned)((n * Q::NPAT + 3) / 4)), dim3(256), lds_main, stream, a.alive, lo,
                           n, recs, a.vk, a.eps, a.label_mode, terms);
        if (ev1 && lo + n >= a.n_cand) (void)hipEventRecord(ev1, stream);
        if (a.label_mode != 0) {
            const size_t lds_x = (size_t)(Q::NCOV + T + Q::LAT + 128 + 16 * Q::PRIME) * sizeof(double);
            static ItalLdsFlags lds_xflags;
            if (lds_x > 48 * 1024) {
                const int rc = ital_raise_lds_limit(reinterpret_cast<const void*>(&qmc_exact_kernel<T>), (int)lds_x, lds_xflags, "ital_score_step");
                if (rc) return rc;
            }
            ITAL_LAUNCH(qmc_exact_kernel<T>, dim3((unsigned)(n * Q::NPAT)), dim3(64), lds_x, stream, a.alive, lo, n, recs, a.vk, a.eps,
                        a.label_mode, terms);
        }
        ITAL_LAUNCH(qmc_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, terms, a.alive, lo, n,
                    Q::NPAT, a.label_mode, a.mi, a.pos_offset, a.gpos, a.sel, part0, nparts, lo + n >= a.n_cand ? 1 : 0);
        part0 += (int)((n + 255) / 256);
        int rc = ital_check_launch("ital_score_step(qmc)");
        if (rc) return rc;
    }
    return 0;
}

template <int T>
static int64_t qmc_cand_doubles() { return Qmc<T>::CAND_DOUBLES; }

static int64_t cand_doubles(int t) {
    switch (t) {
        case 3: return qmc_cand_doubles<3>();
        case 4: return qmc_cand_doubles<4>();
        case 5: return qmc_cand_doubles<5>();
        case 6: return qmc_cand_doubles<6>();
        case 7: return qmc_cand_doubles<7>();
        case 8: return qmc_cand_doubles<8>();
    }
    return 0;
}

}  // namespace ital

using namespace ital;

extern "C" int64_t ital_score_workspace(int t, int64_t n_cand) {
    if (t < 3 || t > ITAL_MAX_T || n_cand <= 0) return 0;
    return cand_doubles(t) * n_cand;
}

// ---- sizes a host needs for the buffers it owns (the prose of ital_hip.h as functions)
extern "C" int ital_record_len(int ldx, int ldw, int kmax) {
    if (ldx < 0 || ldw < 0 || kmax < 0) return 0;
    return ITAL_REC_HEADER + ldx + ldw + kmax;
}

extern "C" int64_t ital_round_workspace(int k, int64_t n_cand, int64_t cap_doubles) {
    if (k > ITAL_MAX_T) k = ITAL_MAX_T;
    const int64_t one_slab = ital_score_workspace(k, n_cand > 0 ? n_cand : 1);     // steps t < k need less per candidate
    if (one_slab == 0) return 0;
    int64_t cap = cap_doubles > 0 ? cap_doubles : one_slab;
    const int64_t least = ital_score_workspace(k, 1);
    if (cap < least) cap = least;
    return one_slab < cap ? one_slab : cap;
}

extern "C" int64_t ital_sel_parts_len(int k, int64_t n_cand, int64_t work_doubles) {
    if (n_cand < 0) n_cand = 0;
    if (k > ITAL_MAX_T) k = ITAL_MAX_T;
    // blocks of the scoring launches of steps 1 .. k: n/256 at t = 1, n/32 at t = 2, n/256 + one (partial) block per slab of
    // the workspace from t = 3 on (launch_qmc); every step re-uses the buffer, so the largest step counts
    int64_t slabs = 0;
    if (k >= 3 && n_cand > 0) {
        const int64_t per = ital_score_workspace(k, 1);
        const int64_t fit = work_doubles / per;
        slabs = fit >= n_cand ? 1 : (fit < 1 ? n_cand : (n_cand + fit - 1) / fit);
    }
    return 3 * (n_cand / 32 + 64 + slabs);
}

// seeds_ready: the generator states of the candidates are already in the workspace (the round driver computed them in the
// launch of the covariance column before).  seeds_only != nullptr: nothing is launched -- the descriptor of the seed
// computation of this step is returned (0), or 1 when the step needs several slabs (then the step seeds itself).
int ital_score_step_internal(const ital_score_desc* d, hipStream_t stream, bool seeds_ready, ital::SeedArgs* seeds_only) {
    if (!d) return ital_fail(-22, "ital_score_step: null descriptor");
    if (seeds_only && (d->n_cand <= 0 || d->t < 3 || d->t > ITAL_MAX_T || !d->work || !d->jump)) return 1;
    if (d->n_cand <= 0) return 0;
    if (d->t < 1 || d->t > ITAL_MAX_T) return ital_fail(-22, "ital_score_step: batch dimension outside 1..ITAL_MAX_T");
    if (d->t > d->batch.kmax) return ital_fail(-22, "ital_score_step: t exceeds the batch capacity");
    ScoreArgs a = {};
    a.t = d->t; a.n_cand = d->n_cand; a.cand = d->cand; a.alive = d->alive; a.mu = d->mu; a.s2 = d->s2; a.C = d->C;
    a.ldc = d->ldc; a.row_offset = d->row_offset; a.pos_offset = d->pos_offset; a.gpos = d->gpos; a.b = d->batch;
    a.noise = d->noise; a.eps = d->eps; a.label_mode = d->label_mode; a.mi = d->mi; a.jump = d->jump;
    a.jumppat = d->jumppat; a.vk = d->vk; a.status = d->status;
    for (int i = 0; i < 6; i++) a.seed[i] = d->seed[i];
    if (d->sel_record) {
        // the selection of this step as the tail of its last scoring launch (what ital_select_fused does separately)
        if (!d->sel_parts || !d->sel_counter || !d->sel_X || !d->sel_xnorm || (d->sel_m > 0 && !d->sel_V))
            return ital_fail(-22, "ital_score_step: fused selection needs sel_parts, sel_counter, sel_X, sel_xnorm, sel_V");
        if (d->sel_m > d->sel_ldw || d->sel_ldx != d->batch.ldx || d->sel_ldw != d->batch.ldw)
            return ital_fail(-22, "ital_score_step: fused selection: batch layout mismatch");
        a.sel.rec = {d->cand, d->pos_offset, d->gpos, d->row_offset, d->sel_rank, 0, 0, d->mu, d->s2, d->sel_X, d->sel_xnorm,
                     d->sel_ldx, d->sel_V, d->sel_ldv, d->sel_m, d->sel_ldw, d->C, d->ldc, d->t - 1, d->batch.kmax, nullptr,
                     d->sel_record, d->status};
        a.sel.slot = d->t - 1;
        a.sel.b = d->batch;
        a.sel.alive = const_cast<uint8_t*>(d->alive);
        a.sel.ret = d->sel_ret;
        a.sel.parts = d->sel_parts;
        a.sel.counter = d->sel_counter;
        a.sel.enabled = 1;
    }
    if (d->t == 1) {
        const int64_t blocks = (d->n_cand + 255) / 256;
        if (a.sel.enabled && 3 * blocks > d->sel_parts_len) return ital_fail(-22, "ital_score_step: sel_parts too small (t = 1)");
        ITAL_LAUNCH(score_t1_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        return ital_check_launch("ital_score_step(t=1)");
    }
    if (d->t == 2) {
        const int64_t blocks = (d->n_cand + 31) / 32;
        if (a.sel.enabled && 3 * blocks > d->sel_parts_len) return ital_fail(-22, "ital_score_step: sel_parts too small (t = 2)");
        ITAL_LAUNCH(score_t2_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        return ital_check_launch("ital_score_step(t=2)");
    }
    if (!d->jump || !d->jumppat || !d->vk) return ital_fail(-22, "ital_score_step: jump tables / generators missing for t >= 3");
    if (!d->work) return ital_fail(-22, "ital_score_step: workspace missing for t >= 3 (see ital_score_workspace)");
    hipEvent_t ev0 = static_cast<hipEvent_t>(d->ev_start), ev1 = static_cast<hipEvent_t>(d->ev_stop);
    switch (d->t) {
        case 3: return launch_qmc<3>(a, d->work, d->work_doubles, d->sel_parts_len, ev0, ev1, stream, seeds_ready, seeds_only);
        case 4: return launch_qmc<4>(a, d->work, d->work_doubles, d->sel_parts_len, ev0, ev1, stream, seeds_ready, seeds_only);
        case 5: return launch_qmc<5>(a, d->work, d->work_doubles, d->sel_parts_len, ev0, ev1, stream, seeds_ready, seeds_only);
        case 6: return launch_qmc<6>(a, d->work, d->work_doubles, d->sel_parts_len, ev0, ev1, stream, seeds_ready, seeds_only);
        case 7: return launch_qmc<7>(a, d->work, d->work_doubles, d->sel_parts_len, ev0, ev1, stream, seeds_ready, seeds_only);
        case 8: return launch_qmc<8>(a, d->work, d->work_doubles, d->sel_parts_len, ev0, ev1, stream, seeds_ready, seeds_only);
    }
    return ital_fail(-22, "ital_score_step: unsupported batch dimension");
}

extern "C" int ital_score_step(const ital_score_desc* d, hipStream_t stream) {
    return ital_score_step_internal(d, stream, false, nullptr);
}
