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
//   t >= 3: Genz MVNDST randomised Korobov lattice (one wave per candidate, lattice points across the 64 lanes,
//           COVSRT variable re-ordering per call by one lane each, FP64-VALU bound).  The lattice shifts replay
//           MVNUNI's stream at the offset the reference's *serial* loop would reach for this (candidate, pattern,
//           call): jump-ahead by 3x3 matrix powers, so the scores - and the argmax - match the reference's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "ital_hip.h"
#include "ital_internal.h"
#include "qmc_common.h"

#ifndef ITAL_QMC_CHUNK_LOG2
#define ITAL_QMC_CHUNK_LOG2(T) 3   // calls per pass of a wave: 8, one lane per (call, shift) when the lattices are drawn (measured: 4 calls
                                   // per pass -- twice the work items -- is 14 % slower at t = 3 and 7 % at t = 4: the per-pass preparation costs more than the finer grid gains)
#endif
#ifndef ITAL_QMC_WAVES
#define ITAL_QMC_WAVES(T) ((T) <= 4 ? 3 : 2)   // waves per SIMD the register allocation aims at (measured)
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
    int64_t pos_offset;     // global list position of local position 0
    ital_batch b;
    double noise, eps;
    int label_mode;         // 0 mean, 1 optimistic, 2 pessimistic
    double* mi;             // [n_cand]
    // MVNUNI replay
    int seed[6];            // generator state at the first call of this greedy step
    const long long* jump;  // [48][18]: transition matrices for 2^b calls (of this step's dimension)
    const long long* jumplane;  // [64][18]: transition matrices for 0..63 shifts (8 shifts = one call)
    int* seeds;             // [n_cand * nsplit][6] generator state at the first call of every work item
    int nsplit;             // work items per candidate (label_mode 0 only; power of two <= NCALLS / CHUNK)
    double* part;           // [n_cand][nsplit] partial sums when nsplit > 1
    const double* vk;       // [t-1] Korobov generators
    int* status;
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
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_cand) return;
    if (!a.alive[p]) return;
    const int row = a.cand[p];
    const double mu = a.mu[row];
    const double su = a.s2[row];               // unclamped: what the simulated update sees (gp.py:334)
    const double sc = fmax(0.0, su);           // clamped: predict_stored(cov_mode='diag') (gp.py:229, ital.py:558)
    const double p_irr = norm_cdf0(mu, sqrt(sc));
    const double w = 1.0 / (su + a.noise);
    const double g = su * w;
    const double s_upd = a.noise * g;
    double mi = 0.0;
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
    if (threadIdx.x < 32) {
        const int64_t pc = (int64_t)blockIdx.x * 32 + threadIdx.x;
        if (pc < a.n_cand && a.alive[pc]) {
            double mi = 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) mi_accumulate(mi, vals[0][4 * threadIdx.x + q], vals[1][4 * threadIdx.x + q], a.eps, a.label_mode);
            a.mi[pc] = mi;
        }
    }
}

// ------------------------------------------------------------------------------------------------ t >= 3
template <int T>
struct Qmc {
    static constexpr int NDIM = T - 1;
    static constexpr int NP = NDIM < 10 ? NDIM : 10;
    static constexpr int PRIME = P_TAB[NP - 1];
    static constexpr int NCOV = T * (T + 1) / 2;
    static constexpr int SLAB_RAW = NCOV + 2 * T;            // packed factor, limits, expected values
    static constexpr int SLAB = SLAB_RAW | 1;                 // odd stride: conflict-free per-lane slabs
    static constexpr int NCALLS = 2 << T;                     // 2 * 2^T
    static constexpr int CHUNK_LOG2 = ITAL_QMC_CHUNK_LOG2(T);
    static constexpr int CHUNK = 1 << CHUNK_LOG2;             // calls prepared per pass (LDS: slab + lattice per call), <= 8
    static constexpr int NCHUNK = NCALLS / CHUNK;             // work items a candidate can be split into
    static constexpr int NCOR = T * (T - 1) / 2;
    // wave-shared candidate area (doubles): pivot, correl, mu0', G, sd', then ints perm
    static constexpr int A_PIVOT = 0, A_COR = A_PIVOT + T, A_MU0 = A_COR + NCOR, A_G = A_MU0 + T, A_SD = A_G + T * T,
                         A_SIZE = A_SD + T;
    static constexpr int LAT = 8 * NDIM * 2;                  // per call: permuted generators + shifts, 8 shifts
    static constexpr int TAILQ = 128 * ITAL_QMC_NH;           // compaction queue of the Phi^-1 tail branch (in place)
    static constexpr int SWAPS = (CHUNK * 8 * NDIM + 1) / 2;  // ints: transposition targets of every (call, shift)
    static constexpr int WAVE_DOUBLES = CHUNK * (SLAB + LAT) + A_SIZE + T + TAILQ + SWAPS + CHUNK;  // + perm, per-call sums
};

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

// MVNUNI state at the first call of every work item (candidate position p, part): the step's seed advanced by
// (rank of p among the live positions) * ncalls + part * ncalls / nsplit calls.  One thread per item: the jump-ahead
// (a 3x3 matrix product mod m per set bit of the offset) costs ~5000 scalar instructions when a wave does it for itself,
// which at 12 waves per CU on one scalar unit was ~10 % of the scorer's time.
__global__ __launch_bounds__(256) void qmc_seed_kernel(ScoreArgs a, int ncalls, int* __restrict__ seeds) {
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = item / a.nsplit;
    if (p >= a.n_cand || !a.alive[p]) return;
    const int part = (int)(item - p * a.nsplit);
    const int64_t gpos = a.pos_offset + p;
    int64_t before = gpos;
    for (int i = 0; i < a.t - 1; i++) before -= (a.b.bgpos[i] < gpos) ? 1 : 0;
    uint64_t calls_before = (uint64_t)before * (uint64_t)ncalls + (uint64_t)(part * (ncalls / a.nsplit));
    MrgState rng = {a.seed[0], a.seed[1], a.seed[2], a.seed[3], a.seed[4], a.seed[5]};
    for (int bit = 0; calls_before != 0; bit++, calls_before >>= 1)
        if (calls_before & 1) mrg_apply(rng, a.jump + bit * 18);
    int* sp = seeds + item * 6;
    sp[0] = rng.x10; sp[1] = rng.x11; sp[2] = rng.x12; sp[3] = rng.x20; sp[4] = rng.x21; sp[5] = rng.x22;
}

template <int T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ITAL_QMC_WAVES(T), ITAL_QMC_WAVES(T)))) void score_qmc_kernel(ScoreArgs a) {
    using Q = Qmc<T>;
    extern __shared__ double lds_all[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // work item of this wave: candidate position p, chunks [part * cpi, (part + 1) * cpi) of its calls.  Splitting the
    // candidates evens out the last scheduling round of the grid (9298 one-candidate waves on 3072 wave slots take as
    // long as 12288 would); the partial sums are combined in a fixed order by score_combine_kernel.
    const int64_t item = (int64_t)blockIdx.x * 4 + wid;
    const int64_t p = item / a.nsplit;
    const int part = (int)(item - p * a.nsplit);
    const int chunk_lo = part * (Q::NCALLS / a.nsplit), chunk_hi = chunk_lo + Q::NCALLS / a.nsplit;
    if (p >= a.n_cand) return;
    if (!a.alive[p]) return;
    double* W = lds_all + (size_t)wid * Q::WAVE_DOUBLES;
    double* slabs = W;
    double* area = W + Q::CHUNK * Q::SLAB;
    double* lats = area + Q::A_SIZE;         // per prepared call: [8][NDIM] generators, then [8][NDIM] shifts
    int* perm = reinterpret_cast<int*>(lats + Q::CHUNK * Q::LAT);  // T ints: natural index held at each sorted slot
    double* tailq = lats + Q::CHUNK * Q::LAT + T;
    int* swaps = reinterpret_cast<int*>(tailq + Q::TAILQ);
    double* vals = tailq + Q::TAILQ + Q::SWAPS;   // lattice sums of the calls of a pass

    const int row = a.cand[p];
    const int64_t gi = a.row_offset + row;

    // ---------------- Phase A: candidate-level quantities (identical in every lane)
    {
        double mean[T], Sg[T][T];
#pragma unroll
        for (int i = 0; i < T - 1; i++) {
            mean[i] = a.b.bmu[i];
#pragma unroll
            for (int j = 0; j < T - 1; j++) Sg[i][j] = a.b.sig[i * a.b.kmax + j];
            const double c = a.C[(int64_t)i * a.ldc + row];
            Sg[i][T - 1] = c;
            Sg[T - 1][i] = c;
        }
        mean[T - 1] = a.mu[row];
        Sg[T - 1][T - 1] = a.s2[row];  // not clamped (gp.py:254)
        double sd[T];
#pragma unroll
        for (int i = 0; i < T; i++) {
            sd[i] = sqrt(Sg[i][i]);
            area[Q::A_PIVOT + i] = -mean[i] / sd[i];
        }
#pragma unroll
        for (int i = 1; i < T; i++)
#pragma unroll
            for (int j = 0; j < i; j++) area[Q::A_COR + i * (i - 1) / 2 + j] = Sg[i][j] / (sd[i] * sd[j]);
        // W = (Sigma + noise I)^-1 through its Cholesky factor, G = I - noise W
        double Lc[T][T];
#pragma unroll
        for (int i = 0; i < T; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) {
                double v = Sg[i][j] + (i == j ? a.noise : 0.0);
#pragma unroll
                for (int q = 0; q < j; q++) v -= Lc[i][q] * Lc[j][q];
                Lc[i][j] = (i == j) ? sqrt(v) : v / Lc[j][j];
            }
        double Li[T][T];  // inverse of the factor (lower)
#pragma unroll
        for (int j = 0; j < T; j++) {
#pragma unroll
            for (int i = 0; i < T; i++) {
                if (i < j) { Li[i][j] = 0; continue; }
                double v = (i == j) ? 1.0 : 0.0;
#pragma unroll
                for (int q = j; q < i; q++) v -= Lc[i][q] * Li[q][j];
                Li[i][j] = v / Lc[i][i];
            }
        }
        double G[T][T];
#pragma unroll
        for (int i = 0; i < T; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) {
                double w = 0;
#pragma unroll
                for (int q = i; q < T; q++) w += Li[q][i] * Li[q][j];
                const double gij = (i == j ? 1.0 : 0.0) - a.noise * w;
                G[i][j] = gij;
                G[j][i] = gij;
            }
#pragma unroll
        for (int i = 0; i < T; i++) {
            double gm = 0;
#pragma unroll
            for (int j = 0; j < T; j++) {
                gm += G[i][j] * mean[j];
                area[Q::A_G + i * T + j] = G[i][j];
            }
            area[Q::A_MU0 + i] = mean[i] - gm;
            area[Q::A_SD + i] = sqrt(a.noise * G[i][i]);
        }
        // sorted-by-data-index order of (batch members, candidate)
        int rank = 0;
#pragma unroll
        for (int i = 0; i < T - 1; i++) rank += (a.b.bidx[i] < gi) ? 1 : 0;
#pragma unroll
        for (int s = 0; s < T; s++) {
            int src;
            if (s < rank) src = a.b.bsort[s];
            else if (s == rank) src = T - 1;
            else src = a.b.bsort[s - 1];
            perm[s] = src;
        }
    }
    // ---------------- stream position of this work item (prepared by qmc_seed_kernel, one thread per item)
    MrgState rng;
    {
        const int* sp = a.seeds + item * 6;
        rng.x10 = __builtin_amdgcn_readfirstlane(sp[0]); rng.x11 = __builtin_amdgcn_readfirstlane(sp[1]);
        rng.x12 = __builtin_amdgcn_readfirstlane(sp[2]); rng.x20 = __builtin_amdgcn_readfirstlane(sp[3]);
        rng.x21 = __builtin_amdgcn_readfirstlane(sp[4]); rng.x22 = __builtin_amdgcn_readfirstlane(sp[5]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    double mi = 0.0;
    double pr_cur = 0.0;
    for (int chunk = chunk_lo; chunk < chunk_hi; chunk += Q::CHUNK) {
        // ---------------- Phase B: lane l prepares call chunk + l (limits, pattern bits, COVSRT) in its LDS slab
        bool sat = false, okc = true;
        unsigned infi = 0;
        if (lane < Q::CHUNK) {
            const int call = chunk + lane;
            const int r = call >> 1;
            double* cov = slabs + lane * Q::SLAB;
            double* lim = cov + Q::NCOV;
            double* y = lim + T;
            if ((call & 1) == 0) {
                for (int i = 0; i < T; i++) {
                    lim[i] = area[Q::A_PIVOT + i];
                    infi |= (unsigned)((r >> (T - 1 - i)) & 1) << i;
                    for (int j = 0; j < i; j++) cov[pidx(i, j)] = area[Q::A_COR + i * (i - 1) / 2 + j];
                    cov[pidx(i, i)] = 1.0;
                }
            } else {
                for (int s = 0; s < T; s++) {
                    const int nat = perm[s];
                    double mu_u = area[Q::A_MU0 + nat];
                    for (int j = 0; j < T; j++) {
                        const double f = ((r >> (T - 1 - j)) & 1) ? 1.0 : -1.0;
                        mu_u += area[Q::A_G + nat * T + j] * f;
                    }
                    const double sds = area[Q::A_SD + nat];
                    lim[s] = -mu_u / sds;
                    infi |= (unsigned)((r >> (T - 1 - nat)) & 1) << s;
                    for (int s2 = 0; s2 < s; s2++) {
                        const int nat2 = perm[s2];
                        cov[pidx(s, s2)] = (a.noise * area[Q::A_G + nat * T + nat2]) / (sds * area[Q::A_SD + nat2]);
                    }
                    cov[pidx(s, s)] = 1.0;
                }
            }
            okc = covsrt<T>(cov, lim, y, infi);
            // integrand identically 1?  every conditional limit stays beyond +-37 for any |y| <= 9
            sat = okc;
            for (int i = 0; i < T; i++) {
                double bound = 0;
                for (int j = 0; j < i; j++) bound += fabs(cov[pidx(i, j)]) * 9.0;
                const bool lower = (infi >> i) & 1u;
                if (lower) { if (!(lim[i] + bound < -37.0)) sat = false; }
                else { if (!(lim[i] - bound > 37.0)) sat = false; }
            }
        }
        // The 8 randomly shifted lattices of every call that is evaluated.  A call draws 8*(2*NDIM-1) uniforms from MVNUNI
        // whether it is evaluated or not: per shift NDIM-1 for DKSMRC's random transposition of the generator vector, then
        // NDIM shifts.  Lane 8c + s jumps to shift s of call c (one matrix product) and replays only that shift's draws;
        // the transpositions, which accumulate from shift to shift, are then applied by the call's own lane.
        {
            const int c = lane >> 3, sft = lane & 7;
            const bool sat_c = __shfl((int)sat, c, 64) != 0;
            if (c < Q::CHUNK && !sat_c) {
                MrgState sti = rng;
                mrg_apply(sti, a.jumplane + lane * 18);
                MrgStateF st = mrg_to_f(sti);
                int* sw = swaps + (c * 8 + sft) * Q::NDIM;
                for (int j = 1; j <= Q::NDIM - 1; j++) {
                    const double u = mrg_next_f(st);
                    sw[j - 1] = (int)(j + u * (Q::NDIM + 1 - j));
                }
                double* L = lats + c * Q::LAT;
                for (int j = 0; j < Q::NDIM; j++) L[8 * Q::NDIM + sft * Q::NDIM + j] = mrg_next_f(st);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < Q::CHUNK && !sat) {
            double* L = lats + lane * Q::LAT;
            for (int j = 0; j < Q::NDIM; j++) L[j] = a.vk[j];
            for (int sft = 0; sft < 8; sft++) {
                double* row = L + sft * Q::NDIM;
                if (sft > 0)
                    for (int j = 0; j < Q::NDIM; j++) row[j] = row[j - Q::NDIM];
                const int* sw = swaps + (lane * 8 + sft) * Q::NDIM;
                for (int j = 1; j <= Q::NDIM - 1; j++) {
                    const int jp = sw[j - 1];
                    const double xt = row[j - 1];
                    row[j - 1] = row[jp - 1];
                    row[jp - 1] = xt;
                }
            }
        }
        mrg_apply(rng, a.jump + Q::CHUNK_LOG2 * 18);   // the wave's base state moves on by CHUNK calls
        if (!okc) atomicOr(a.status, 2);  // singular conditional covariance: not supported by this kernel
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---------------- Phase C: the wave evaluates the calls of this chunk one after the other
        for (int cl = 0; cl < Q::CHUNK; cl++) {
            // wave-uniform copies (SGPR) of the preparing lane's flags: keeps the generator arithmetic on the scalar unit
            const bool sat_c = __builtin_amdgcn_readlane((int)sat, cl) != 0;
            const unsigned infi_c = (unsigned)__builtin_amdgcn_readlane((int)infi, cl);
            if (!sat_c) {
                const double* lat = lats + cl * Q::LAT;
                // per-call constants out of the preparing lane's slab, as wave-uniform (scalar) values
                const double* cov = slabs + cl * Q::SLAB;
                double cf[Q::NCOR > 0 ? Q::NCOR : 1], lm[T];
#pragma unroll
                for (int i = 0; i < T; i++) {
                    lm[i] = uniform_f64(cov[Q::NCOV + i]);
#pragma unroll
                    for (int j = 0; j < i; j++) cf[i * (i - 1) / 2 + j] = uniform_f64(cov[pidx(i, j)]);
                }
                const double acc = qmc_lane_sum<T>(lat, cf, lm, infi_c, tailq, lane);
                const double tot = wave_sum(acc);
                if (lane == 0) vals[cl] = tot;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the calls' values enter the sum in call order (prior probability, then the one after the simulated update)
        for (int cl = 0; cl < Q::CHUNK; cl++) {
            const int call = chunk + cl;
            const bool sat_c = __builtin_amdgcn_readlane((int)sat, cl) != 0;
            const double value = sat_c ? 1.0 : vals[cl] / (16.0 * Q::PRIME);
            if ((call & 1) == 0) {
                pr_cur = value;
            } else {
                mi_accumulate(mi, pr_cur, value, a.eps, a.label_mode);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane == 0) {
        if (a.nsplit == 1) a.mi[p] = mi;
        else a.part[p * a.nsplit + part] = mi;
    }
}

// mi[p] = sum of the partial sums of position p, in part order (deterministic).
__global__ __launch_bounds__(256) void score_combine_kernel(const double* __restrict__ part, const uint8_t* __restrict__ alive,
                                                            int64_t n_cand, int nsplit, double* __restrict__ mi) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_cand || !alive[p]) return;
    double s = 0.0;
    for (int j = 0; j < nsplit; j++) s += part[p * nsplit + j];
    mi[p] = s;
}

template <int T>
static int launch_qmc(const ScoreArgs& a, hipStream_t stream) {
    using Q = Qmc<T>;
    const size_t lds = (size_t)4 * Q::WAVE_DOUBLES * sizeof(double);
    const int64_t blocks = (a.n_cand * a.nsplit + 3) / 4;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&score_qmc_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ital_fail(-12, "score_qmc: cannot raise the dynamic LDS limit");
        attr_done = true;
    }
    const int64_t items = a.n_cand * a.nsplit;
    hipLaunchKernelGGL(qmc_seed_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, a, Q::NCALLS, a.seeds);
    int rc0 = ital_check_launch("ital_score_step(seeds)");
    if (rc0) return rc0;
    hipLaunchKernelGGL(score_qmc_kernel<T>, dim3((unsigned)blocks), dim3(256), lds, stream, a);
    int rc = ital_check_launch("ital_score_step(qmc)");
    if (rc || a.nsplit == 1) return rc;
    hipLaunchKernelGGL(score_combine_kernel, dim3((unsigned)((a.n_cand + 255) / 256)), dim3(256), 0, stream, a.part, a.alive,
                       a.n_cand, a.nsplit, a.mi);
    return ital_check_launch("ital_score_step(combine)");
}

}  // namespace ital

using namespace ital;

extern "C" int ital_score_step(const ital_score_desc* d, hipStream_t stream) {
    if (!d) return ital_fail(-22, "ital_score_step: null descriptor");
    if (d->n_cand <= 0) return 0;
    if (d->t < 1 || d->t > ITAL_MAX_T) return ital_fail(-22, "ital_score_step: batch dimension outside 1..ITAL_MAX_T");
    if (d->t > d->batch.kmax) return ital_fail(-22, "ital_score_step: t exceeds the batch capacity");
    ScoreArgs a = {};
    a.t = d->t; a.n_cand = d->n_cand; a.cand = d->cand; a.alive = d->alive; a.mu = d->mu; a.s2 = d->s2; a.C = d->C;
    a.ldc = d->ldc; a.row_offset = d->row_offset; a.pos_offset = d->pos_offset; a.b = d->batch; a.noise = d->noise;
    a.eps = d->eps; a.label_mode = d->label_mode; a.mi = d->mi; a.jump = d->jump; a.jumplane = d->jumplane; a.vk = d->vk; a.status = d->status;
    for (int i = 0; i < 6; i++) a.seed[i] = d->seed[i];
    if (d->t == 1) {
        hipLaunchKernelGGL(score_t1_kernel, dim3((unsigned)((d->n_cand + 255) / 256)), dim3(256), 0, stream, a);
        return ital_check_launch("ital_score_step(t=1)");
    }
    if (d->t == 2) {
        hipLaunchKernelGGL(score_t2_kernel, dim3((unsigned)((d->n_cand + 31) / 32)), dim3(256), 0, stream, a);
        return ital_check_launch("ital_score_step(t=2)");
    }
    if (!d->jump || !d->jumplane || !d->vk) return ital_fail(-22, "ital_score_step: jump tables / generators missing for t >= 3");
    if (!d->seeds) return ital_fail(-22, "ital_score_step: seeds buffer missing for t >= 3");
    a.seeds = d->seeds;
    a.nsplit = 1;
    a.part = nullptr;
    if (d->split > 1 && d->partial && d->label_mode == 0) {
        const int max_split = (2 << d->t) >> ITAL_QMC_CHUNK_LOG2(d->t);   // NCALLS / CHUNK
        int ns = 1;
        while (ns * 2 <= d->split && ns * 2 <= max_split) ns *= 2;
        a.nsplit = ns;
        a.part = d->partial;
    }
    switch (d->t) {
        case 3: return launch_qmc<3>(a, stream);
        case 4: return launch_qmc<4>(a, stream);
        case 5: return launch_qmc<5>(a, stream);
        case 6: return launch_qmc<6>(a, stream);
        case 7: return launch_qmc<7>(a, stream);
        case 8: return launch_qmc<8>(a, stream);
    }
    return ital_fail(-22, "ital_score_step: unsupported batch dimension");
}
