// General mutual-information scorer: every combination of the reference's options that the perfect-user fast path
// (score.hip) does not cover --
//   * user models (reference ital/ital.py:300-342 `fb_iter`, :453-481 `likelihood`): perfect, "motivated"
//     (label_prob >= 1, mistake_prob > 0: feedback in {-1,+1}^t) and general (feedback in {-1,0,+1}^t without the
//     all-zero configuration; zero feedbacks are dropped from the simulated update, retrieval_base.py:192-193);
//   * the change-estimation subset (ital.py:227-275 `_call_iter_sub`, :504-529, :541-582): the orthant
//     probabilities run over U = subset + picks (+ candidate), of which only the picks and the candidate are
//     enumerated; the others keep the sign of their predictive mean;
//   * label_estimation mean / optimistic / pessimistic (ital.py:210-219, `_call_iter_all` only).
//
// Per candidate (one wave) the reference's call sequence is reproduced in its serial order:
//   for pattern r (itertools.product order):  prior call(s);  for feedback f (product order, all-zero skipped): updated call
// "prior" = prob_rel over U in natural order (plain mode), or prob_rel over the enumerated variables followed by
// prob_rel over all of U (subset mode); "updated" = prob_rel over U sorted by data index under the closed-form
// posterior after the simulated update with the non-zero feedbacks F:
//     W = (Sigma_FF + s I)^-1, g = W (f - mu_F),
//     mu'_F = f - s g,  mu'_a = mu_a + Sigma_aF g,
//     Sigma'_FF = s (I - s W),  Sigma'_aF = s Sigma_aF W,  Sigma'_ab = Sigma_ab - Sigma_aF W Sigma_Fb   (s = noise)
// (cancellation-free on the fed-back block; equal to gp.updated_prediction's (m+|F|)-dimensional inverse).
// Calls of dimension 1 / 2 are closed forms (norm.cdf / Genz BVU) and draw nothing; dimension >= 3 is Genz's
// MVNDST lattice rule with the MVNUNI stream replayed at the offset the serial reference reaches (jump-ahead by the
// number of uniforms the preceding candidates and calls consume).  Lane l prepares call l of a chunk (closed-form
// update, COVSRT) in its own LDS slab; the wave then evaluates the chunk's calls one after the other with the lattice
// points spread over the lanes (4 chains per lane, runtime dimension <= NMAX with uniform early exits).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "device_math.h"
#include "ital_hip.h"
#include "ital_internal.h"
#ifndef ITAL_GEN_SHORT_JUMP
// measured in this kernel: the jump with one reduction per matrix row (device_math.h) makes the t = 3 instantiation 5 %
// faster and the t = 4 one 3 % slower (register allocation around the out-of-line lattice routine); the noisy-user round
// as a whole is 2 % faster with a reduction per product
#define mrg_apply mrg_apply_each
#endif
#include "qmc_common.h"

#ifndef ITAL_GEN_NH12
#define ITAL_GEN_NH12 1   // lattice items per lane and round of the runtime-dimension evaluator up to 12 dimensions
#endif
#ifndef ITAL_GEN_NOINLINE
#define ITAL_GEN_NOINLINE __attribute__((noinline))   // keeps the preparation's registers out of the evaluation loop's budget
#endif
#ifndef ITAL_GEN_HOTK
#define ITAL_GEN_HOTK 1   // exp / log coefficients of the compile-time-dimension evaluator as vector-register operands (HotK:
                          // 197 -> 20 spilled VGPRs, 385 -> 193 spilled SGPRs in the t = 4 instantiation; noisy-user step 61.8 -> 59.1 ms)
#endif
#ifndef ITAL_GEN_BIG_NCB
// chains per lane and round of the compile-time evaluator for 7 .. 16 variables, at two waves per SIMD: as many as the
// 256 registers hold without spilling much -- the tail branch of Phi^-1 then runs on fuller waves.  Measured per step
// (40 000 x 512, monte_carlo_num_rel = 1): 7 variables 57 -> 44 ms, 8: 122 -> 78, 9: 235 -> 160 (these ran two chains at
// three waves per SIMD before), 10: 318 -> 282 (four chains), 11-14: three chains (-6 % .. -1 %; four lose 10 % at 12),
// 15 / 16: two (three spill 63 / 76 registers: +3 % / +20 %)
// Round 3: with the coefficients of the Phi^-1 tail branch materialised in place (device_math.h lit_s) every instantiation
// lost ~70 registers and all scratch; chains per lane raised to what 256 registers hold now -- six at 7, 8 variables, five
// at 9, 10, four at 11, 12, three beyond (40 000 x 512, k = 16, monte_carlo_num_rel = 1: 5.80 -> 5.53 s per round;
// profiles/r3_general_variants.txt)
// ... and then traded against occupancy where three waves per SIMD (168 registers) hold at least two chains without
// scratch: per step at 40 000 x 512 (profiles/r3_general_variants.txt, second block) 8 variables 67.1 ms (six chains, two
// waves) -> 59.7 (four chains, three waves), 10: 274.8 -> 250.8 (three), 13: 714.6 -> 691.9 (two), 14: 834.8 -> 816.2;
// 11, 12 and 15, 16 stay at two waves (four / three chains: three waves would spill or lose)
#define ITAL_GEN_BIG_NCB(T) ((T) <= 8 ? 4 : (T) <= 10 ? 3 : (T) <= 12 ? 4 : (T) <= 14 ? 2 : 3)
#endif
#ifndef ITAL_GEN_BIG_HOTK
#define ITAL_GEN_BIG_HOTK 1
#endif
#ifndef ITAL_GEN_BIG_COEF
#define ITAL_GEN_BIG_COEF HotKE6     // six exp coefficients in registers, four in place (the logarithm of the tail branch takes
                                    // its own in place anyway): what the three-wave instantiations need to stay without scratch
#endif
#ifndef ITAL_GEN_FIXED_NH
// lattice items per lane and round of the pipeline's evaluator for 3 .. 6 variables: three (six chains) at 5 and 6
// variables, which then run at two waves per SIMD (per step at 40 000 x 512: 13.6 -> 11.3 ms, 26.5 -> 20.5 ms; four items
// lose again); two at 3 and 4, three waves per SIMD
// Round 3 (registers freed by lit_s): three items = six chains at every dimension, see ITAL_GEN_MAIN_WAVES
#define ITAL_GEN_FIXED_NH(T) 3
#endif
#ifndef ITAL_GEN_ONE_TRIP
// dimensions whose lattice-sum launch runs one call per wave (grid = capacity of the list; waves beyond its length leave
// at once) instead of a fixed grid of waves striding over the list: all compile-time evaluators.  Around the evaluator the
// striding loop cost registers in every instantiation (t = 4: 41 doubles spilled, 3.5 GB of scratch traffic per launch);
// without it: noisy-user t = 4 step 42 -> 34.8 ms, monte_carlo_num_rel steps at 7 .. 16 variables -7 % .. -20 %
#define ITAL_GEN_ONE_TRIP(T) ((T) >= 3)
#endif
#ifndef ITAL_GEN_TAILQ
#define ITAL_GEN_TAILQ 384     // doubles per wave of the Phi^-1 tail queue of the pipeline's lattice sums: up to 6 chains per lane
#endif
#ifndef ITAL_GEN_MAIN_WAVES
// waves per SIMD the lattice-sum kernels aim at: four up to 4 variables (120 registers with six chains), three at 5 and 6,
// two beyond (noisy-user round 41.0 -> 36.5 ms with six chains at these occupancies; four chains at four waves: 38.1)
#define ITAL_GEN_MAIN_WAVES(T) ((T) > 0 && (T) < 5 ? 4 : (((T) >= 5 && (T) <= 10) || (T) == 13 || (T) == 14 ? 3 : 2))
#endif
#ifndef ITAL_GEN_TFIX_MAX
#define ITAL_GEN_TFIX_MAX 16   // largest dimension the pipeline takes (plain mode)
#endif
#ifndef ITAL_GEN_PREP_SPLIT
#define ITAL_GEN_PREP_SPLIT 4   // preparation waves per candidate in the pipeline
#endif
#ifndef ITAL_GEN_PIPELINE
#define ITAL_GEN_PIPELINE 1   // plain mode, 3 .. 16 variables: prepare / lattice sums / combine as three kernels on streams of their own
#endif
#ifndef ITAL_GEN_EARLY
#define ITAL_GEN_EARLY 1   // decide saturated calls from the standardised limits, before COVSRT
#endif

namespace ital {

constexpr int GN = ITAL_GENERIC_MAX_DIM;  // largest orthant dimension
constexpr int GR = ITAL_GENERIC_MAX_REL;  // largest number of enumerated variables

struct GArgs {
    ital_gscore_desc d;
    int chunk;         // calls prepared per pass (<= 64)
    int stride;        // doubles per preparing lane (odd)
    int slab;          // of which the COVSRT slab (packed factor, limits, expected values), then the update scratch,
    int lat;           // then (at this offset) the call's 8 shifted lattices
    int ldS;           // leading dimension of the joint covariance in LDS (largest |U|)
    int master;        // clip_cov: offset of the lane's copy of the standardised problem (0: clip_cov off)
    int wave_doubles;  // LDS doubles per wave
};

// ------------------------------------------------------------------------------------------------ COVSRT, runtime n
__device__ void rcswp_n(int n, int p, int q, double* cov, double* lim, unsigned& infi) {
    double tmp = lim[p]; lim[p] = lim[q]; lim[q] = tmp;
    unsigned bp = (infi >> p) & 1u, bq = (infi >> q) & 1u;
    infi = (infi & ~((1u << p) | (1u << q))) | (bq << p) | (bp << q);
    tmp = cov[pidx(p, p)]; cov[pidx(p, p)] = cov[pidx(q, q)]; cov[pidx(q, q)] = tmp;
    for (int j = 0; j < p; j++) { tmp = cov[pidx(p, j)]; cov[pidx(p, j)] = cov[pidx(q, j)]; cov[pidx(q, j)] = tmp; }
    for (int i = p + 1; i < q; i++) { tmp = cov[pidx(i, p)]; cov[pidx(i, p)] = cov[pidx(q, i)]; cov[pidx(q, i)] = tmp; }
    for (int i = q + 1; i < n; i++) { tmp = cov[pidx(i, p)]; cov[pidx(i, p)] = cov[pidx(i, q)]; cov[pidx(i, q)] = tmp; }
}

__device__ bool covsrt_n(int n, double* cov, double* lim, double* y, unsigned& infi) {
    const double SQTWPI = 2.506628274631001, EPS = 1e-10;
    bool ok = true;
    for (int i = 0; i < n; i++) {
        double dmin = 0, emin = 1, zmin = 0, cvdiag = 0;
        int jmin = i;
        for (int j = i; j < n; j++) {
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
        if (jmin > i) rcswp_n(n, i, jmin, cov, lim, infi);
        cov[pidx(i, i)] = cvdiag;
        if (cvdiag > 0) {
            const double rdiag = fast_div(1.0, cvdiag);   // cvdiag in (1e-5, 1]: well scaled
            for (int l = i + 1; l < n; l++) {
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
            // zero diagonal (linearly dependent variable): Genz's COVSRT expresses the row through the last earlier
            // variable it depends on and moves it right behind that variable; MVNDFN then intersects the limits
            for (int l = i + 1; l < n; l++) cov[pidx(l, i)] = 0;
            for (int j = i - 1; j >= 0; j--) {
                const double cij = cov[pidx(i, j)];
                if (fabs(cij) > EPS) {
                    lim[i] = lim[i] / cij;
                    if (cij < 0) infi ^= 1u << i;
                    for (int l = 0; l <= j; l++) cov[pidx(i, l)] = cov[pidx(i, l)] / cij;
                    for (int l = j + 1; l <= i - 1; l++) {
                        if (cov[pidx(l, j + 1)] > 0) {
                            for (int k = i - 1; k >= l; k--) {
                                for (int m = 0; m <= k; m++) {
                                    const double tmp = cov[pidx(k, m)];
                                    cov[pidx(k, m)] = cov[pidx(k + 1, m)];
                                    cov[pidx(k + 1, m)] = tmp;
                                }
                                const double tl = lim[k]; lim[k] = lim[k + 1]; lim[k + 1] = tl;
                                const unsigned bk = (infi >> k) & 1u, bk1 = (infi >> (k + 1)) & 1u;
                                infi = (infi & ~((1u << k) | (1u << (k + 1)))) | (bk1 << k) | (bk << (k + 1));
                            }
                            break;
                        }
                    }
                    break;
                }
                cov[pidx(i, j)] = 0;
            }
            y[i] = 0;
        }
    }
    return ok;
}

// After COVSRT: which rows close a group of MVNDFN (a row whose successor has a positive diagonal, or the last row),
// and the factor re-packed so that column (closing row of group g) carries the coefficient of group g -- the
// evaluator can then index its y registers by row.  Bit i of the result = row i closes a group.
__device__ unsigned group_layout(int n, double* cov) {
    unsigned closes = 0;
    int ik = 0;                    // groups closed before the current row
    unsigned long long crow0 = 0, crow1 = 0;   // closing row of each group, 5 bits each (12 groups per word)
    for (int i = 0; i < n; i++) {
        // columns >= ik of row i are ignored by MVNDFN; re-pack columns g < ik to column crow[g]
        const int nk = ik < i ? ik : i;
        for (int j = nk; j < i; j++) cov[pidx(i, j)] = 0;
        for (int g = nk - 1; g >= 0; g--) {
            const int r = g < 12 ? (int)((crow0 >> (5 * g)) & 31ull) : (int)((crow1 >> (5 * (g - 12))) & 31ull);
            if (r != g) {
                cov[pidx(i, r)] = cov[pidx(i, g)];
                cov[pidx(i, g)] = 0;
            }
        }
        const bool close = (i == n - 1) || (cov[pidx(i + 1, ik + 1)] > 0);
        if (close) {
            closes |= 1u << i;
            if (ik < 12) crow0 |= (unsigned long long)i << (5 * ik);
            else crow1 |= (unsigned long long)i << (5 * (ik - 12));
            ik++;
        }
    }
    return closes;
}

// ------------------------------------------------------------------------------------------------ call list
enum { K_PRIOR = 0, K_PRIOR_SUB = 1, K_PRIOR_FULL = 2, K_UPDATED = 3 };

struct CallInfo {
    int kind;
    unsigned pat;     // pattern bits, enumerated variable v at bit (nr-1-v)
    unsigned fnz;     // enumerated variables with non-zero feedback (bit v)
    unsigned fpos;    // ... whose feedback is +1 (bit v)
    double weight;    // likelihood of the feedback given the pattern (ital.py:472-481), 1 for a perfect user
};

__device__ __forceinline__ int pow3(int n) {
    int r = 1;
    for (int i = 0; i < n; i++) r *= 3;
    return r;
}

enum { K_SKIP = 4 };

// Call `call` of candidate position p.  cpp = calls per pattern (npre prior calls + nfb feedback configurations).
__device__ CallInfo decode_call(const ital_gscore_desc& d, int64_t p, int call, int cpp, int npre, int nr, int npat) {
    CallInfo c;
    const bool subset = d.subset_mode != 0;
    const int pi = call / cpp, s = call - pi * cpp;
    c.pat = d.mc_rel > 0 ? d.rel_samples[p * d.mc_rel + pi] : (unsigned)pi;
    c.fnz = 0; c.fpos = 0; c.weight = 1.0;
    if (s < npre) {
        c.kind = subset ? (s == 0 ? K_PRIOR_SUB : K_PRIOR_FULL) : K_PRIOR;
        return c;
    }
    c.kind = K_UPDATED;
    int f = s - npre;
    if (d.fb_mode == 0) {
        c.fnz = (1u << nr) - 1u;
        for (int v = 0; v < nr; v++) c.fpos |= ((c.pat >> (nr - 1 - v)) & 1u) << v;
        return c;
    }
    if (d.mc_fb > 0) {
        const unsigned w = d.fb_samples[(p * npat + pi) * d.mc_fb + f];
        c.fnz = w & 0xffffu;
        c.fpos = w >> 16;
        c.weight = 1.0 / d.mc_fb;
        if (c.fnz == 0) c.kind = K_SKIP;   // all-zero feedback: no call at all (ital.py:201)
        return c;
    }
    if (d.fb_mode == 1) {
        c.fnz = (1u << nr) - 1u;
        for (int v = 0; v < nr; v++) c.fpos |= (((unsigned)f >> (nr - 1 - v)) & 1u) << v;
    } else {
        const int zero = (pow3(nr) - 1) / 2;
        if (f >= zero) f++;
        int rem = f;
        for (int v = nr - 1; v >= 0; v--) {   // last variable = least significant base-3 digit
            const int dgt = rem % 3;
            rem /= 3;
            if (dgt != 1) c.fnz |= 1u << v;
            if (dgt == 2) c.fpos |= 1u << v;
        }
    }
    const double lp = d.label_prob, mp = d.mistake_prob;
    double w = 1.0;
    for (int v = 0; v < nr; v++) {
        const bool r = (c.pat >> (nr - 1 - v)) & 1u;
        if (!((c.fnz >> v) & 1u)) w *= 1.0 - lp;
        else if ((((c.fpos >> v) & 1u) != 0) == r) w *= lp * (1.0 - mp);
        else w *= lp * mp;
    }
    c.weight = w;
    return c;
}

// position of the q-th set bit helpers are avoided: loops run over the set bits directly
#define FOR_BITS(mask, u, q) for (unsigned _m = (mask), q = 0, u = 0; _m && ((u = __builtin_ctz(_m)), true); _m &= _m - 1, q++)

struct Prep {
    int n;
    unsigned infi;
    unsigned closes; // rows that close a group of MVNDFN (all rows unless the covariance is singular)
    int flags;       // 1 closed form (value valid), 2 integrand == 1, 4 integrand == 0, 16 skipped, 32 clip_cov groups
    int ng;          // clip_cov: number of independent groups (flags & 32)
    int gdraws;      // clip_cov: uniforms the groups' calls consume in total
    double value;
};

// Early decision without COVSRT.  Whatever order COVSRT picks, the conditional limit of variable a is
// (lim_a - sum_j c_aj y_j) / c_aa with sum_j c_aj^2 + c_aa^2 = 1 and |y_j| <= 9, i.e. it stays beyond +-37 once
// |lim_a| > 37 + 9 sqrt(n - 1): one variable on the empty side makes every lattice point contribute exactly 0, all
// variables on the full side make every point contribute exactly 1 -- the values the full path returns.
// Returns 4 (== 0), 2 (== 1) or 0 (undecided).
__device__ int early_decision(int n, const double* lim, unsigned infi) {
    const double thr = 37.0 + 9.0 * sqrt((double)(n - 1));
    bool all_full = true, any_empty = false;
    for (int a = 0; a < n; a++) {
        const bool lower = (infi >> a) & 1u;
        const double l = lower ? lim[a] : -lim[a];     // interval [l, inf) in the variable's own direction
        if (l > thr) any_empty = true;
        if (!(l < -thr)) all_full = false;
    }
    return any_empty ? 4 : (all_full ? 2 : 0);
}

// Standardised problem (limits lim, packed correlations cov with unit diagonal, n >= 2) -> closed form (n = 2) or the
// COVSRT-ed slab with its saturation verdict.
__device__ void finish_call(int n, double* cov, double* lim, double* y, Prep& out) {
    if (n == 2) {
        out.value = bvn_orthant(lim[0], lim[1], out.infi & 1u, (out.infi >> 1) & 1u, cov[pidx(1, 0)]);
        out.flags = 1;
        return;
    }
    covsrt_n(n, cov, lim, y, out.infi);
    out.closes = group_layout(n, cov);
    bool sat1 = true, sat0 = false;
    for (int i = 0; i < n; i++) {
        double bound = 0;
        for (int j = 0; j < i; j++) bound += fabs(cov[pidx(i, j)]) * 9.0;
        const bool lower = (out.infi >> i) & 1u;
        if (lower) {
            if (!(lim[i] + bound < -37.0)) sat1 = false;
            if (lim[i] - bound > 37.0) sat0 = true;
        } else {
            if (!(lim[i] - bound > 37.0)) sat1 = false;
            if (lim[i] + bound < -37.0) sat0 = true;
        }
    }
    if (sat0) out.flags = 4;
    else if (sat1) out.flags = 2;
}

// clip_cov (reference ital/ital.py:386-429, :590-616): connected components of |corr| > clip over the n variables, in
// group_cov's order -- seeds ascending, members in breadth-first layers, each layer ascending.  Writes the member order
// and the group boundaries; returns the number of groups.
__device__ int clip_groups(int n, const double* cor, double clip, int* adj, int* gorder, int* gstart) {
    for (int a = 0; a < n; a++) {
        unsigned m = 0;
        for (int b = 0; b < n; b++) {
            const double c = a == b ? 1.0 : (a > b ? cor[pidx(a, b)] : cor[pidx(b, a)]);
            if (fabs(c) > clip) m |= 1u << b;
        }
        adj[a] = (int)m;
    }
    unsigned left = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
    int ng = 0, pos = 0;
    while (left) {
        gstart[ng] = pos;
        unsigned newm = (unsigned)adj[__builtin_ctz(left)];
        unsigned grp = 0;
        while (newm) {
            for (unsigned m = newm; m; m &= m - 1) gorder[pos++] = __builtin_ctz(m);
            grp |= newm;
            left &= ~newm;
            unsigned reach = 0;
            for (unsigned m = grp; m; m &= m - 1) reach |= (unsigned)adj[__builtin_ctz(m)];
            newm = reach & left;
        }
        ng++;
    }
    gstart[ng] = pos;
    return ng;
}

// Sub-problem of group g of a clip_cov call: singleton -> norm.cdf, pair -> BVU, larger -> standardised slab + COVSRT.
__device__ ITAL_GEN_NOINLINE Prep build_group(int g, const double* mlim, const double* mcor, unsigned infi_full, const int* gorder,
                            const int* gstart, double* slab) {
    Prep out;
    out.flags = 0; out.value = 0; out.infi = 0; out.closes = 0; out.ng = 0; out.gdraws = 0;
    const int g0 = gstart[g], n = gstart[g + 1] - g0;
    out.n = n;
    if (n == 1) {
        const int u = gorder[g0];
        const double q = ndtr(mlim[u]);                       // norm.cdf(0, mean, sd), lim = -mean / sd
        out.value = ((infi_full >> u) & 1u) ? 1.0 - q : q;
        out.flags = 1;
        return out;
    }
    double* cov = slab;
    double* lim = slab + n * (n + 1) / 2;
    double* y = lim + n;
    for (int a = 0; a < n; a++) {
        const int ua = gorder[g0 + a];
        lim[a] = mlim[ua];
        out.infi |= ((infi_full >> ua) & 1u) << a;
        for (int b = 0; b < a; b++) {
            const int ub = gorder[g0 + b];
            cov[pidx(a, b)] = ua > ub ? mcor[pidx(ua, ub)] : mcor[pidx(ub, ua)];
        }
        cov[pidx(a, a)] = 1.0;
    }
    if (ITAL_GEN_EARLY && n >= 3) {
        const int e = early_decision(n, lim, out.infi);
        if (e) { out.flags = e; return out; }
    }
    finish_call(n, cov, lim, y, out);
    return out;
}

// Prepares one call in the lane's slab: cov (packed, n(n+1)/2), lim (n), y (n); scratch fs for the update.
template <bool CLIP>
__device__ ITAL_GEN_NOINLINE Prep prepare_call(const ital_gscore_desc& d, const CallInfo& ci, int nU, int nr, int ldS, const double* muU,
                             const double* SigU, const int* usort, const int* ipos, bool clamp_prior, double* slab,
                             double* fs, double* master) {
    Prep out;
    out.flags = 0; out.value = 0; out.infi = 0; out.closes = 0; out.ng = 0; out.gdraws = 0;
    const bool subset = d.subset_mode != 0;
    const int n = ci.kind == K_PRIOR_SUB ? nr : nU;
    out.n = n;
    double* cov = slab;
    double* lim = slab + n * (n + 1) / 2;
    double* y = lim + n;
    // sign of every variable of U: enumerated ones from the pattern, the rest from the predictive mean (ital.py:238)
    unsigned relU = 0;
    if (subset)
        for (int u = 0; u < nU; u++) relU |= (muU[u] > 0 ? 1u : 0u) << u;
    for (int v = 0; v < nr; v++) {
        const unsigned bit = (ci.pat >> (nr - 1 - v)) & 1u;
        relU = (relU & ~(1u << ipos[v])) | (bit << ipos[v]);
    }
    // feedback set F as positions of U
    unsigned Fm = 0, Fp = 0;
    if (ci.kind == K_UPDATED)
        for (int v = 0; v < nr; v++)
            if ((ci.fnz >> v) & 1u) {
                Fm |= 1u << ipos[v];
                if ((ci.fpos >> v) & 1u) Fp |= 1u << ipos[v];
            }
    const int nf = __builtin_popcount(Fm);
    double* M = fs;               // nf x nf
    double* dg = fs + nf * nf;    // diagonal of W
    double* gv = dg + nf;         // W (f - mu_F)
    const double s = d.noise;
    if (nf > 0) {
        FOR_BITS(Fm, ua, qa) {
            FOR_BITS(Fm, ub, qb) {
                if (qb > qa) break;
                M[qa * nf + qb] = SigU[ua * ldS + ub] + (qa == qb ? s : 0.0);
            }
        }
        for (int i = 0; i < nf; i++)
            for (int j = 0; j <= i; j++) {
                double v = M[i * nf + j];
                for (int q = 0; q < j; q++) v -= M[i * nf + q] * M[j * nf + q];
                M[i * nf + j] = (i == j) ? sqrt(v) : v / M[j * nf + j];
            }
        // in-place inverse of the lower factor (column by column, the columns to the right still hold L)
        for (int j = 0; j < nf; j++) {
            M[j * nf + j] = 1.0 / M[j * nf + j];
            for (int i = j + 1; i < nf; i++) {
                double sm = 0;
                for (int q = j; q < i; q++) sm += M[i * nf + q] * M[q * nf + j];
                M[i * nf + j] = -sm / M[i * nf + i];
            }
        }
        // W = X^T X: off-diagonal into the upper triangle, diagonal aside
        for (int j = 0; j < nf; j++) {
            for (int i = 0; i < j; i++) {
                double w = 0;
                for (int k = j; k < nf; k++) w += M[k * nf + i] * M[k * nf + j];
                M[i * nf + j] = w;
            }
        }
        for (int i = 0; i < nf; i++) {
            double w = 0;
            for (int k = i; k < nf; k++) w += M[k * nf + i] * M[k * nf + i];
            dg[i] = w;
        }
        FOR_BITS(Fm, ua, qa) {
            double acc = 0;
            FOR_BITS(Fm, ub, qb) {
                const double w = qa == qb ? dg[qa] : (qa < qb ? M[qa * nf + qb] : M[qb * nf + qa]);
                const double fb = ((Fp >> ub) & 1u) ? 1.0 : -1.0;
                acc += w * (fb - muU[ub]);
            }
            gv[qa] = acc;
        }
    }
    auto Wat = [&](unsigned qa, unsigned qb) -> double {
        return qa == qb ? dg[qa] : (qa < qb ? M[qa * nf + qb] : M[qb * nf + qa]);
    };
    auto upos_of = [&](int a) -> int {
        return ci.kind == K_PRIOR_SUB ? ipos[a] : (ci.kind == K_UPDATED ? usort[a] : a);
    };
    // posterior mean (into lim) and covariance (packed) of the call's variables, then standardise.  First the means
    // and the diagonal only: they decide, for most updated calls, that the integrand is identically 0 or 1.
    auto cov_entry = [&](int ua, bool aF, unsigned qa, int ub, bool bF, unsigned qb) -> double {
        if (nf == 0) return SigU[ua * ldS + ub];
        if (aF && bF) return s * ((qa == qb ? 1.0 : 0.0) - s * Wat(qa, qb));
        if (bF) {
            double acc = 0;
            FOR_BITS(Fm, uf, q) acc += SigU[ua * ldS + uf] * Wat(q, qb);
            return s * acc;
        }
        if (aF) {
            double acc = 0;
            FOR_BITS(Fm, uf, q) acc += SigU[ub * ldS + uf] * Wat(q, qa);
            return s * acc;
        }
        double acc = 0;
        FOR_BITS(Fm, uf, q) {
            double inner = 0;
            FOR_BITS(Fm, ug, q2) inner += Wat(q, q2) * SigU[ug * ldS + ub];
            acc += SigU[ua * ldS + uf] * inner;
        }
        return SigU[ua * ldS + ub] - acc;
    };
    for (int a = 0; a < n; a++) {
        const int ua = upos_of(a);
        const bool aF = (Fm >> ua) & 1u;
        const unsigned qa = __builtin_popcount(Fm & ((1u << ua) - 1u));
        double mean;
        if (aF) {
            mean = (((Fp >> ua) & 1u) ? 1.0 : -1.0) - s * gv[qa];
        } else {
            mean = muU[ua];
            FOR_BITS(Fm, uf, q) mean += SigU[ua * ldS + uf] * gv[q];
        }
        lim[a] = mean;
        out.infi |= ((relU >> ua) & 1u) << a;
        cov[pidx(a, a)] = cov_entry(ua, aF, qa, ua, aF, qa);
    }
    if (clamp_prior && ci.kind != K_UPDATED && n == 1) cov[0] = fmax(0.0, cov[0]);  // predict_stored 'diag' (gp.py:229)
    for (int a = 0; a < n; a++) y[a] = sqrt(cov[pidx(a, a)]);  // standard deviations, for now
    if (n == 1) {
        const double p_irr = norm_cdf0(lim[0], y[0]);      // prob_rel, ital.py:364-369
        out.value = (out.infi & 1u) ? 1.0 - p_irr : p_irr;
        out.flags = 1;
        return out;
    }
    for (int a = 0; a < n; a++) lim[a] = -lim[a] / y[a];
    const bool clip_mode = CLIP && master != nullptr && n > 5;        // prob_rel -> _grouped_prob_rel (ital.py:360-362)
    if (ITAL_GEN_EARLY && n >= 3 && !clip_mode) {
        const int e = early_decision(n, lim, out.infi);
        if (e) { out.flags = e; return out; }
    }
    for (int a = 0; a < n; a++) {
        const int ua = upos_of(a);
        const bool aF = (Fm >> ua) & 1u;
        const unsigned qa = __builtin_popcount(Fm & ((1u << ua) - 1u));
        for (int b = 0; b < a; b++) {
            const int ub = upos_of(b);
            const bool bF = (Fm >> ub) & 1u;
            const unsigned qb = __builtin_popcount(Fm & ((1u << ub) - 1u));
            cov[pidx(a, b)] = cov_entry(ua, aF, qa, ub, bF, qb) / (y[a] * y[b]);
        }
    }
    for (int a = 0; a < n; a++) cov[pidx(a, a)] = 1.0;
    if (CLIP && clip_mode) {
        double* mlim = master;
        double* mcor = master + n;
        int* adj = reinterpret_cast<int*>(mcor + n * (n + 1) / 2);
        int* gorder = adj + n;
        int* gstart = gorder + n;
        const int ng = clip_groups(n, cov, d.clip_cov, adj, gorder, gstart);
        if (ng > 1) {
            for (int a = 0; a < n; a++) mlim[a] = lim[a];
            for (int e = 0; e < n * (n + 1) / 2; e++) mcor[e] = cov[e];
            int draws = 0;
            for (int g = 0; g < ng; g++) {
                const int sz = gstart[g + 1] - gstart[g];
                if (sz >= 3) draws += 8 * (2 * (sz - 1) - 1);
            }
            out.flags = 32;
            out.ng = ng;
            out.gdraws = draws;
            return out;
        }
    }
    finish_call(n, cov, lim, y, out);
    return out;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// One MVNDST pass (8 shifted lattices, antithetic pairs) for a prepared call of dimension n <= NMAX; every lane runs
// NH lattice items per round, each with its antithetic partner (2*NH independent chains).
template <int NMAX, int NH>
__device__ double qmc_eval(int n, const double* __restrict__ slab, unsigned infi, unsigned closes,
                           const double* __restrict__ lat, int lane, double* __restrict__ tailq) {
    constexpr int NC = 2 * NH;
    const int ndim = n - 1;
    const int prime = P_TAB[(ndim < 10 ? ndim : 10) - 1];
    const double* cf = slab;
    const double* lm = slab + n * (n + 1) / 2;
    const int items = 8 * prime;
    double acc = 0.0;
    for (int base = 0; base < items; base += 64 * NH) {
        double yy[NC][NMAX - 1], ff[NC], ai[NC], bi[NC];
        bool dead[NC];
        int kk[NH], so[NH];
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int item = base + 64 * h + lane;
            const bool ok = item < items;
            const int it = ok ? item : 0;
            const int sft = it / prime;
            kk[h] = it - sft * prime + 1;
            so[h] = sft * ndim;
            ff[2 * h] = ff[2 * h + 1] = 1.0;
            dead[2 * h] = dead[2 * h + 1] = !ok;
        }
#pragma unroll
        for (int c = 0; c < NC; c++) { ai[c] = 0; bi[c] = 0; }
        bool infa = false, infb = false;   // wave-uniform: the open group has a lower / an upper limit (MVNDFN)
        int ik = 0;                        // groups closed so far = lattice coordinate of the open group
#pragma unroll
        for (int i = 0; i < NMAX; i++) {
            if (i < n) {   // uniform; no `break`: the body holds convergent wave operations and must stay unrollable
                const bool lower = (infi >> i) & 1u;
                const bool close = (closes >> i) & 1u;
                const bool last = i == n - 1;
                const double lmi = lm[i];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    double sc = 0;
#pragma unroll
                    for (int j = 0; j < i; j++) sc = fma(cf[pidx(i, j)], yy[c][j], sc);
                    const double z = lmi - sc;
                    if (lower) ai[c] = infa ? fmax(ai[c], z) : z;
                    else bi[c] = infb ? fmin(bi[c], z) : z;
                }
                if (lower) infa = true; else infb = true;
                if (close) {
                    double xh[NH];
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        xh[h] = 0;
                        if (!last) {
                            const double v = kk[h] * lat[so[h] + ik] + lat[8 * ndim + so[h] + ik];
                            const double fr = v - floor(v);
                            xh[h] = fabs(2 * fr - 1);
                        }
                    }
                    double pin[NC];
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const double dd = infa ? mvn_phi(ai[c]) : 0.0;
                        const double ee = infb ? mvn_phi(bi[c]) : 1.0;
                        const double w = ee - dd;
                        dead[c] = dead[c] || !(w > 0);
                        ff[c] *= w;
                        const double x = (c & 1) ? 1 - xh[c >> 1] : xh[c >> 1];
                        pin[c] = fma(x, w, dd);
                    }
                    if (!last) {
                        double outv[NC];
                        phinv_wave<NC>(pin, outv, tailq, lane);
#pragma unroll
                        for (int c = 0; c < NC; c++)
                            if (i < NMAX - 1) yy[c][i < NMAX - 1 ? i : 0] = outv[c];
                    }
                    infa = false; infb = false;
                    ik++;
                } else {
#pragma unroll
                    for (int c = 0; c < NC; c++)
                        if (i < NMAX - 1) yy[c][i < NMAX - 1 ? i : 0] = 0.0;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) acc += dead[c] ? 0.0 : ff[c];
    }
    return wave_sum(acc) / (16.0 * prime);
}

// One MVNDST pass for a call of compile-time dimension T whose rows all close their own group (no linearly dependent
// variable): the evaluator of the perfect-user fast path (score.hip) -- factor and limits as wave-uniform scalars, fully
// unrolled, NHF lattice items x antithetic partner per lane.
template <int T, int NHF = 2, bool FL = false>
__device__ __forceinline__ double qmc_eval_fixed_inl(const double* __restrict__ slab, unsigned infi,
                                                     const double* __restrict__ lat, int lane, double* __restrict__ tailq) {
    constexpr int NDIM = T - 1, NCOV = T * (T + 1) / 2, NCOR = T * (T - 1) / 2;
    constexpr int PRIME = P_TAB[(NDIM < 10 ? NDIM : 10) - 1];
    double cf[NCOR > 0 ? NCOR : 1], lm[T];
#pragma unroll
    for (int i = 0; i < T; i++) {
        lm[i] = uniform_f64(slab[NCOV + i]);
#pragma unroll
        for (int j = 0; j < i; j++) cf[i * (i - 1) / 2 + j] = uniform_f64(slab[pidx(i, j)]);
    }
#if ITAL_GEN_HOTK
    HotK kk;
    kk.load();
    const double acc = qmc_lane_sum<T, HotK, NHF, FL>(lat, cf, lm, infi, tailq, lane, kk);
#else
    const double acc = qmc_lane_sum<T, LitK, NHF, FL>(lat, cf, lm, infi, tailq, lane);
#endif
    return wave_sum(acc) / (16.0 * PRIME);
}

template <int T>
__device__ double qmc_eval_fixed(const double* __restrict__ slab, unsigned infi, const double* __restrict__ lat, int lane,
                                 double* __restrict__ tailq) {
    return qmc_eval_fixed_inl<T>(slab, infi, lat, lane, tailq);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The 8 randomly shifted lattices of one call of dimension n, generated by the calling lane from the generator state
// `base` advanced by `before` uniforms.
__device__ ITAL_GEN_NOINLINE void make_lattice(const ital_gscore_desc& d, const MrgState& base, unsigned before, int n, double* L) {
    MrgState sti = base;
    for (int bit = 0; before != 0; bit++, before >>= 1)
        if (before & 1u) mrg_apply(sti, d.jump1 + bit * 18);
    MrgStateF st = mrg_to_f(sti);
    const int ndim = n - 1;
    for (int j = 0; j < ndim; j++) L[j] = d.vk[n * GN + j];
    for (int sft = 0; sft < 8; sft++) {
        double* row = L + sft * ndim;
        if (sft > 0)
            for (int j = 0; j < ndim; j++) row[j] = row[j - ndim];
        for (int j = 1; j <= ndim - 1; j++) {
            const double u = mrg_next_f(st);
            const int jp = (int)(j + u * (ndim + 1 - j));
            const double xt = row[j - 1];
            row[j - 1] = row[jp - 1];
            row[jp - 1] = xt;
        }
        for (int j = 0; j < ndim; j++) L[8 * ndim + sft * ndim + j] = mrg_next_f(st);
    }
}

#ifndef ITAL_GEN_WAVES
#define ITAL_GEN_WAVES(TFIX) ((TFIX) > 0 ? 3 : 2)   // measured: compile-time-dimension evaluator 3 waves per SIMD, runtime one 2 (spills)
#endif
template <int NMAX, int NH, int TFIX, bool CLIP>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(ITAL_GEN_WAVES(TFIX), ITAL_GEN_WAVES(TFIX)))) void score_generic_kernel(GArgs a) {
    extern __shared__ double lds_all[];
    const ital_gscore_desc& d = a.d;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t p = (int64_t)blockIdx.x * 2 + wid;
    if (p >= d.n_cand) return;
    if (!d.alive[p]) return;
    double* W = lds_all + (size_t)wid * a.wave_doubles;
    double* muU = W;
    const int ldS = a.ldS;                       // = largest |U| of this launch
    double* SigU = muU + ldS;
    int* usort = reinterpret_cast<int*>(SigU + ldS * ldS);
    int* ipos = usort + GN;
    double* tailq = SigU + ldS * ldS + (GN + GR + 1) / 2;
    double* slabs = tailq + ITAL_GEN_TAILQ;

    const int row = d.cand[p];
    const int64_t gi = d.row_offset + row;
    const int nE = d.nE;
    const bool subset = d.subset_mode != 0;
    int epos = -1;
    for (int e = 0; e < nE; e++)
        if (d.E_idx[e] == gi) epos = e;
    if (!subset) epos = -1;
    const int nU = epos >= 0 ? nE : nE + 1;
    const int cpos = epos >= 0 ? epos : nE;
    const int nr = d.n_picks + 1;

    // ---------------- Phase A: joint mean / covariance of U = E (+ candidate) in the wave's LDS area
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
    // ---------------- stream position of this candidate in the reference's serial order
    MrgState rng = {d.seed[0], d.seed[1], d.seed[2], d.seed[3], d.seed[4], d.seed[5]};
    {
        const int64_t gpos = d.gpos ? d.gpos[p] : d.pos_offset + p;
        int64_t before = gpos, n_in = 0;
        for (int i = 0; i < d.n_dead; i++) before -= (d.dead_pos[i] < gpos) ? 1 : 0;
        for (int i = 0; i < d.n_in; i++) n_in += (d.in_pos[i] < gpos) ? 1 : 0;
        uint64_t off = (uint64_t)(before - n_in) * (uint64_t)d.draws_out + (uint64_t)n_in * (uint64_t)d.draws_in;
        if (d.draw_off) off = (uint64_t)d.draw_off[p];
        for (int bit = 0; off != 0; bit++, off >>= 1)
            if (off & 1) mrg_apply(rng, d.jump1 + bit * 18);
    }
    wave_sync();

    const int npat = d.mc_rel > 0 ? d.mc_rel : (1 << nr);
    const int npre = subset ? 2 : 1;
    const bool entropy = d.fb_mode == 3;   // batch entropy: prior probabilities only (baseline_methods.py:270-287)
    const int nfb = entropy ? 0 : (d.fb_mode == 0 ? 1 : (d.mc_fb > 0 ? d.mc_fb : (d.fb_mode == 1 ? (1 << nr) : pow3(nr) - 1)));
    const int cpp = npre + nfb;
    const int total = npat * cpp;
    const bool clamp_prior = !subset && nr == 1;   // first greedy step: predict_stored(cov_mode='diag') (ital.py:558)

    double mi = 0.0, pr_cur = 0.0, logpr_cur = 0.0;
    unsigned long long pairs = 0;   // (Phi, Phi^-1) pairs of the lattice sums this wave evaluates (bench.py roofline)
    int64_t cand_draws = 0;     // uniforms this candidate's calls consume (reported by the counting pass)
    for (int chunk0 = 0; chunk0 < total; chunk0 += a.chunk) {
        // ---------------- Phase B: lane l prepares call chunk0 + l
        Prep pp;
        pp.n = 0; pp.infi = 0; pp.flags = 0; pp.value = 0; pp.closes = 0; pp.ng = 0; pp.gdraws = 0;
        if (lane < a.chunk && chunk0 + lane < total) {
            const CallInfo ci = decode_call(d, p, chunk0 + lane, cpp, npre, nr, npat);
            double* slab = slabs + (size_t)lane * a.stride;
            if (ci.kind == K_SKIP) { pp.flags = 16; }
            else pp = prepare_call<CLIP>(d, ci, nU, nr, ldS, muU, SigU, usort, ipos, clamp_prior, slab, slab + a.slab,
                                             (CLIP && a.master) ? slab + a.master : nullptr);
        }
        // lattices of the calls that are evaluated, generated lane-parallel: every dimension >= 3 call (evaluated or
        // saturated) takes 8*(2*NDIM-1) uniforms from MVNUNI; lane l jumps ahead by what the calls before it in this chunk
        // consume, the wave's base state by the chunk's total
        {
            const bool draws_any = pp.n >= 3 && !(pp.flags & (1 | 16 | 32));
            const int my_draws = (pp.flags & 32) ? pp.gdraws : (draws_any ? 8 * (2 * (pp.n - 1) - 1) : 0);
            int incl = my_draws;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            const int total_draws = __builtin_amdgcn_readlane(incl, 63);
            cand_draws += total_draws;
            if (d.draw_count) {      // counting pass: only the stream consumption is wanted
                wave_sync();
                continue;
            }
            const unsigned call_base = (unsigned)(incl - my_draws);
            if (draws_any && !(pp.flags & 6))
                make_lattice(d, rng, call_base, pp.n, slabs + (size_t)lane * a.stride + a.lat);
            // clip_cov: calls that fall apart into independent groups (ital.py:413-429) are evaluated group by group --
            // pass g prepares group g of every such call in the call's slab, the wave evaluates the ones that need the
            // lattice rule, and the product of the group probabilities turns the call into a closed-form one
            if (CLIP && __any((pp.flags & 32) != 0)) {
                int maxg = (pp.flags & 32) ? pp.ng : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const int o = __shfl_xor(maxg, off, 64);
                    maxg = o > maxg ? o : maxg;
                }
                double gprod = 1.0;
                unsigned gdone = 0;
                for (int g = 0; g < maxg; g++) {
                    Prep gp;
                    gp.n = 0; gp.infi = 0; gp.flags = 64; gp.value = 1.0; gp.closes = 0; gp.ng = 0; gp.gdraws = 0;
                    if ((pp.flags & 32) && g < pp.ng) {
                        double* slab = slabs + (size_t)lane * a.stride;
                        const double* mlim = slab + a.master;
                        const double* mcor = mlim + pp.n;
                        const int* gorder = reinterpret_cast<const int*>(mcor + pp.n * (pp.n + 1) / 2) + pp.n;
                        gp = build_group(g, mlim, mcor, pp.infi, gorder, gorder + pp.n, slab);
                        if (gp.n >= 3) {
                            if (!(gp.flags & 7)) make_lattice(d, rng, call_base + gdone, gp.n, slab + a.lat);
                            gdone += 8 * (2 * (gp.n - 1) - 1);
                        }
                    }
                    wave_sync();
                    double gval = (gp.flags & 1) ? gp.value : ((gp.flags & 2) ? 1.0 : ((gp.flags & 4) ? 0.0 : 1.0));
                    for (int cl = 0; cl < a.chunk; cl++) {
                        const int fl_c = __builtin_amdgcn_readlane(gp.flags, cl);
                        if (fl_c != 0) continue;     // inactive, closed form or saturated
                        const int n_c = __builtin_amdgcn_readlane(gp.n, cl);
                        const unsigned infi_c = (unsigned)__builtin_amdgcn_readlane((int)gp.infi, cl);
                        const unsigned closes_c = (unsigned)__builtin_amdgcn_readlane((int)gp.closes, cl);
                        const double* slab_c = slabs + (size_t)cl * a.stride;
                        const double v = qmc_eval<NMAX, NH>(n_c, slab_c, infi_c, closes_c, slab_c + a.lat, lane, tailq);
                        pairs += 16ull * P_TAB[(n_c - 1 < 10 ? n_c - 1 : 10) - 1] * (n_c - 1);
                        if (lane == cl) gval = v;
                    }
                    if (!(gp.flags & 64)) gprod *= gval;
                    wave_sync();
                }
                if (pp.flags & 32) {
                    pp.value = gprod;
                    pp.flags = 1;
                }
            }
            unsigned adv = (unsigned)total_draws;
            for (int bit = 0; adv != 0; bit++, adv >>= 1)
                if (adv & 1u) mrg_apply(rng, d.jump1 + bit * 18);
        }
        wave_sync();
        // ---------------- Phase C
        for (int cl = 0; cl < a.chunk && chunk0 + cl < total; cl++) {
            const int n_c = __builtin_amdgcn_readlane(pp.n, cl);
            const int fl_c = __builtin_amdgcn_readlane(pp.flags, cl);
            const unsigned infi_c = (unsigned)__builtin_amdgcn_readlane((int)pp.infi, cl);
            const unsigned closes_c = (unsigned)__builtin_amdgcn_readlane((int)pp.closes, cl);
            double value;
            if (fl_c & 16) continue;   // skipped all-zero feedback sample
            if (fl_c & 1) {
                value = readlane_f64(pp.value, cl);
            } else if (fl_c & 6) {
                value = (fl_c & 2) ? 1.0 : 0.0;
            } else {
                const double* slab_c = slabs + (size_t)cl * a.stride;
                if (TFIX > 0 && n_c == TFIX && closes_c == (1u << (TFIX > 0 ? TFIX : 1)) - 1u)
                    value = qmc_eval_fixed<(TFIX > 0 ? TFIX : 3)>(slab_c, infi_c, slab_c + a.lat, lane, tailq);
                else
                    value = qmc_eval<NMAX, NH>(n_c, slab_c, infi_c, closes_c, slab_c + a.lat, lane, tailq);
                pairs += 16ull * P_TAB[(n_c - 1 < 10 ? n_c - 1 : 10) - 1] * (n_c - 1);
            }
            const CallInfo ci = decode_call(d, p, chunk0 + cl, cpp, npre, nr, npat);
            if (entropy) {
                if (nr == 1) {
                    // single_entropy (baseline_methods.py:263-267): P(irrelevant) clamped to [1e-8, 1 - 1e-8]
                    if (ci.pat == 0) {
                        const double q = fmax(1e-8, fmin(1.0 - 1e-8, value));
                        mi = q * log(q) + (1.0 - q) * log(1.0 - q);
                    }
                } else if (value > 1e-12) {
                    mi += value * log(value);
                }
            } else if (ci.kind == K_PRIOR) {
                pr_cur = value;
                logpr_cur = log(value + d.eps);
            } else if (ci.kind == K_PRIOR_SUB) {
                pr_cur = value;
            } else if (ci.kind == K_PRIOR_FULL) {
                logpr_cur = log(value + d.eps);
            } else {
                double cur = (log(value + d.eps) - logpr_cur) * ci.weight;
                if (!subset && d.label_mode == 1) {
                    if (cur > mi) mi = cur;
                } else if (!subset && d.label_mode == 2) {
                    if (mi == 0 || cur < mi) mi = cur;
                } else {
                    mi += d.mc_rel > 0 ? cur : cur * pr_cur;   // sampled patterns are not weighted (ital.py:216-218)
                }
            }
        }
        wave_sync();
    }
    if (d.draw_count) {
        if (lane == 0) d.draw_count[p] = cand_draws;
        return;
    }
    if (d.mc_rel > 0) mi /= d.mc_rel;   // ital.py:221-222
    if (entropy) mi = -mi;
    if (lane == 0) d.mi[p] = mi;
    if (lane == 0 && d.pair_count) atomicAdd(d.pair_count, pairs);
}

// ------------------------------------------------------------------------------------------------ three-kernel pipeline
// Plain mode with a compile-time evaluator (no subset, no clip_cov, 3 .. 16 variables: the noisy user models, the entropy
// baseline, the Monte-Carlo pattern switch up to batches of 16).  The monolithic kernel above prepares, evaluates and accumulates inside one wave
// per candidate; at the noisy-user benchmark size it fills half of the vector issue slots (59 ms per t = 4 launch for 33 ms
// of lattice sums at the perfect-user kernel's rate).  Here the step is three kernels over slabs of candidates, the prepared
// calls travelling through a workspace in HBM:
//   gen_prep_kernel     wave per candidate, lane per call   decode + closed-form update + early verdict + COVSRT + lattices;
//                                                           calls that need a lattice sum are appended to the slab's list
//   gen_main_kernel<T>  waves striding over that list       the lattice sums (FP64-VALU bound, the perfect-user evaluator)
//   gen_combine_kernel  wave per candidate                  the terms in the reference's order -> mi
// The preparation is bound by the latency of its LDS-resident per-call matrices, the lattice sums by vector issue: the two
// run on streams of their own, the preparation of slab s + 1 and the combine of slab s - 1 under the lattice sums of slab s
// (double-buffered workspace).
// meta[call] = (flags | n << 8 | infi << 16 | closes << 40, value): value is written by the preparation (closed forms,
// saturated calls) or by the lattice-sum kernel.
struct GPipe {
    int64_t slab_lo, slab_n;   // candidate positions [slab_lo, slab_lo + slab_n) of this launch
    int total;                 // calls per candidate
    int R;                     // doubles per record: packed factor + limits (the evaluator's slab), then the lattices
    int lat;                   // offset of the lattices inside a record
    double* meta;              // [slab_n][total][2]
    double* recs;              // [slab_n][total][R]
    unsigned int* list;        // [slab_n * total] indices (candidate of the slab * total + call) of the calls to integrate:
                               // regular ones from the front, the ones with linearly dependent variables from the back
    unsigned int* count;       // [2] entries from the front / from the back
    int nsplit;                // waves a candidate's calls are spread over in the preparation (1 when the stream offsets of the
                               // calls depend on the data: skipped all-zero feedback samples)
};

__device__ __forceinline__ long long pack_meta(const Prep& pp) {
    return (long long)(pp.flags & 0xff) | ((long long)(pp.n & 0xff) << 8) | ((long long)(pp.infi & 0xffffffu) << 16) |
           ((long long)(pp.closes & 0xffffffu) << 40);
}

// The 8 lattices of a call with the generator vector in a scratch area `gen` (LDS, n - 1 doubles), streamed out to `L`
// (write-only: straight into the call's record in HBM).
__device__ ITAL_GEN_NOINLINE void make_lattice_stream(const ital_gscore_desc& d, const MrgState& base, unsigned before, int n,
                                                      double* gen, double* __restrict__ L) {
    MrgState sti = base;
    for (int bit = 0; before != 0; bit++, before >>= 1)
        if (before & 1u) mrg_apply(sti, d.jump1 + bit * 18);
    MrgStateF st = mrg_to_f(sti);
    const int ndim = n - 1;
    for (int j = 0; j < ndim; j++) gen[j] = d.vk[n * GN + j];
    for (int sft = 0; sft < 8; sft++) {
        for (int j = 1; j <= ndim - 1; j++) {
            const double u = mrg_next_f(st);
            const int jp = (int)(j + u * (ndim + 1 - j));
            const double xt = gen[j - 1];
            gen[j - 1] = gen[jp - 1];
            gen[jp - 1] = xt;
        }
        for (int j = 0; j < ndim; j++) L[sft * ndim + j] = gen[j];
        for (int j = 0; j < ndim; j++) L[8 * ndim + sft * ndim + j] = mrg_next_f(st);
    }
}

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
    const int nU = nE + 1;            // plain mode: U = batch so far + candidate
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
        int rank = 0;
        for (int e = 0; e < nE; e++) rank += (d.E_idx[e] < gi) ? 1 : 0;
        for (int sidx = 0; sidx < nU; sidx++)
            usort[sidx] = sidx < rank ? d.E_sort[sidx] : (sidx == rank ? nE : d.E_sort[sidx - 1]);
        for (int v = 0; v < nr; v++) ipos[v] = v < d.n_picks ? d.pick_pos[v] : nE;
    }
    const int npat = d.mc_rel > 0 ? d.mc_rel : (1 << nr);
    const bool entropy = d.fb_mode == 3;
    const int nfb = entropy ? 0 : (d.fb_mode == 0 ? 1 : (d.mc_fb > 0 ? d.mc_fb : (d.fb_mode == 1 ? (1 << nr) : pow3(nr) - 1)));
    const int cpp = 1 + nfb;
    const int total = npat * cpp;
    // this wave's share of the candidate's calls: whole passes of a.chunk calls
    const int npass = (total + a.chunk - 1) / a.chunk;
    const int pass_lo = (int)((int64_t)npass * part / g.nsplit), pass_hi = (int)((int64_t)npass * (part + 1) / g.nsplit);
    // stream position of this candidate in the reference's serial order, then of the share's first call (with more than one
    // share every call draws the same 8 (2 (nU - 1) - 1) uniforms: no skipped samples)
    MrgState rng = {d.seed[0], d.seed[1], d.seed[2], d.seed[3], d.seed[4], d.seed[5]};
    {
        const int64_t gpos = d.gpos ? d.gpos[p] : d.pos_offset + p;
        int64_t before = gpos;
        for (int q = 0; q < d.n_dead; q++) before -= (d.dead_pos[q] < gpos) ? 1 : 0;
        uint64_t off = (uint64_t)before * (uint64_t)d.draws_out;
        if (d.draw_off) off = (uint64_t)d.draw_off[p];
        off += (uint64_t)pass_lo * (uint64_t)a.chunk * (uint64_t)(nU >= 3 ? 8 * (2 * (nU - 1) - 1) : 0);
        for (int bit = 0; off != 0; bit++, off >>= 1)
            if (off & 1) mrg_apply(rng, d.jump1 + bit * 18);
    }
    wave_sync();
    const bool clamp_prior = nr == 1;
    double* meta = g.meta + (size_t)i * total * 2;
    double* recs = g.recs + (size_t)i * total * g.R;
    for (int chunk0 = pass_lo * a.chunk; chunk0 < pass_hi * a.chunk && chunk0 < total; chunk0 += a.chunk) {
        Prep pp;
        pp.n = 0; pp.infi = 0; pp.flags = 16; pp.value = 0; pp.closes = 0; pp.ng = 0; pp.gdraws = 0;
        const int call = chunk0 + lane;
        const bool mine = lane < a.chunk && call < total;
        double* slab = slabs + (size_t)lane * a.stride;
        if (mine) {
            const CallInfo ci = decode_call(d, p, call, cpp, 1, nr, npat);
            if (ci.kind != K_SKIP)
                pp = prepare_call<false>(d, ci, nU, nr, ldS, muU, SigU, usort, ipos, clamp_prior, slab, slab + a.slab, nullptr);
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
                double* rec = recs + (size_t)call * g.R;
                const int ns = pp.n * (pp.n + 1) / 2 + pp.n;
                for (int q = 0; q < ns; q++) rec[q] = slab[q];
                make_lattice_stream(d, rng, (unsigned)(incl - my_draws), pp.n, slab, rec + g.lat);   // the slab is free now
                const unsigned int id = (unsigned int)(i * total + call);
                if (regular) g.list[lbase + (unsigned int)__popcll(em & ((1ull << lane) - 1ull))] = id;
                else g.list[(unsigned int)(g.slab_n * total) - 1u - cbase - (unsigned int)__popcll(cm & ((1ull << lane) - 1ull))] = id;
            }
            double value = pp.value;
            if (!(pp.flags & 1) && (pp.flags & 6)) value = (pp.flags & 2) ? 1.0 : 0.0;
            meta[2 * call] = __longlong_as_double(pack_meta(pp));
            meta[2 * call + 1] = value;
        }
        unsigned adv = (unsigned)total_draws;
        for (int bit = 0; adv != 0; bit++, adv >>= 1)
            if (adv & 1u) mrg_apply(rng, d.jump1 + bit * 18);
        wave_sync();
    }
}

// The lattice sums of the list of calls to integrate; its length is only known on the device.  T > 0: the regular calls (T
// variables, every row closes its own group) with the compile-time evaluator, one call per wave in a grid that covers the
// capacity of the list (ITAL_GEN_ONE_TRIP); T == 0: the calls with linearly dependent variables, from the back of the
// list, with the runtime evaluator in a small fixed grid of waves that stride over them.
template <int T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ITAL_GEN_MAIN_WAVES(T), ITAL_GEN_MAIN_WAVES(T)))) void gen_main_kernel(
    GPipe g, unsigned long long* pair_count) {
    extern __shared__ double lds_all[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* rec = lds_all + (size_t)wid * (g.R + ITAL_GEN_TAILQ);
    double* tailq = rec + g.R;
    const unsigned int count = g.count[T > 0 ? 0 : 1];
    const unsigned int last = (unsigned int)(g.slab_n * g.total) - 1u;
    const unsigned int nwaves = gridDim.x * 4;
    unsigned long long pairs = 0;
    auto integrate = [&](unsigned int e) {
        const unsigned int item = g.list[T > 0 ? e : last - e];
        double* meta = g.meta + (size_t)item * 2;
        const long long m = __double_as_longlong(uniform_f64(meta[0]));
        const int n = (int)((m >> 8) & 0xff);
        const unsigned infi = (unsigned)((m >> 16) & 0xffffffu);
        const double* src = g.recs + (size_t)item * g.R;
        const int ns = n * (n + 1) / 2 + n;
        // the compile-time evaluators take every variable as bounded above (ITAL_QMC_FLIP, qmc_common.h): the signs go in here
        const unsigned fl = (T > 0 && ITAL_QMC_FLIP) ? infi : 0u;
        for (int q = lane; q < ns; q += 64) {
            double v = src[q];
            if (fl) {
                int row = q - n * (n + 1) / 2, col = row;                        // a limit
                if (row < 0) {                                                    // packed lower triangle with diagonal
                    row = 0;
                    while ((row + 1) * (row + 2) / 2 <= q) row++;
                    col = q - row * (row + 1) / 2;
                    if (((fl >> row) ^ (fl >> col)) & 1u) v = -v;
                } else if ((fl >> row) & 1u) {
                    v = -v;
                }
            }
            rec[q] = v;
        }
        for (int q = lane; q < 16 * (n - 1); q += 64) {
            double v = src[g.lat + q];
            if (fl && q >= 8 * (n - 1) && ((fl >> (q % (n - 1))) & 1u)) v += 0.5;   // the shifts of a negated variable
            rec[g.lat + q] = v;
        }
        constexpr bool FL = T > 0 && ITAL_QMC_FLIP != 0;
        const unsigned infi_e = infi;          // FL: the variables that entered negated
        wave_sync();
        double value;
        if (T >= 7) {
            constexpr int TB = T >= 7 ? T : 7, NDIMB = TB - 1;
#if ITAL_GEN_BIG_HOTK
            ITAL_GEN_BIG_COEF kk;      // exp coefficients as vector-register operands (device_math.h), as in the perfect-user kernel
            kk.load();
            value = wave_sum(qmc_lane_sum_big<TB, ITAL_GEN_BIG_NCB(TB), ITAL_GEN_BIG_COEF, FL>(rec + g.lat, rec, infi_e, tailq, lane, kk)) /
                    (16.0 * P_TAB[(NDIMB < 10 ? NDIMB : 10) - 1]);
#else
            value = wave_sum(qmc_lane_sum_big<TB, ITAL_GEN_BIG_NCB(TB), LitK, FL>(rec + g.lat, rec, infi_e, tailq, lane)) /
                    (16.0 * P_TAB[(NDIMB < 10 ? NDIMB : 10) - 1]);
#endif
        } else if (T > 0) {
            constexpr int TF = T > 0 && T < 7 ? T : 3;
            value = qmc_eval_fixed_inl<TF, ITAL_GEN_FIXED_NH(TF), FL>(rec, infi_e, rec + g.lat, lane, tailq);
        }
        else value = qmc_eval<ITAL_GENERIC_MAX_DIM, 1>(n, rec, infi, (unsigned)((m >> 40) & 0xffffffu), rec + g.lat, lane, tailq);
        if (lane == 0) meta[1] = value;
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
        pairs = (blockIdx.x == 0 && wid == 0) ? (unsigned long long)count * (16ull * P_TAB[(ND < 10 ? ND : 10) - 1] * ND) : 0ull;
    } else {
        for (unsigned int e = blockIdx.x * 4 + wid; e < count; e += nwaves) integrate(e);
    }
    if (lane == 0 && pair_count && pairs) atomicAdd(pair_count, pairs);
}

// Wave per candidate: the lanes form the terms of 64 calls at a time, lane 0's order-preserving fold adds them up exactly
// as the reference's loop does (ital.py:207-222).
__global__ __launch_bounds__(256) void gen_combine_kernel(GArgs a, GPipe g) {
    const ital_gscore_desc& d = a.d;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= g.slab_n) return;
    const int64_t p = g.slab_lo + i;
    if (!d.alive[p]) return;
    const int nr = d.n_picks + 1;
    const int npat = d.mc_rel > 0 ? d.mc_rel : (1 << nr);
    const bool entropy = d.fb_mode == 3;
    const int nfb = entropy ? 0 : (d.fb_mode == 0 ? 1 : (d.mc_fb > 0 ? d.mc_fb : (d.fb_mode == 1 ? (1 << nr) : pow3(nr) - 1)));
    const int cpp = 1 + nfb;
    const int total = npat * cpp;
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
                const CallInfo ci = decode_call(d, p, call, cpp, 1, nr, npat);
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
                    const double pr = meta[2 * (call / cpp) * cpp + 1];       // the pattern's prior probability
                    const double cur = (log(value + d.eps) - log(pr + d.eps)) * ci.weight;
                    term = (d.label_mode != 0 || d.mc_rel > 0) ? cur : cur * pr;   // sampled patterns are not weighted
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
            else if (d.label_mode == 1) { if (t_l > mi) mi = t_l; }
            else if (d.label_mode == 2) { if (mi == 0 || t_l < mi) mi = t_l; }
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

// Streams and events of the pipeline (one set per device of the process, created on first use).
struct PipeStreams {
    hipStream_t prep, main, comb;
    hipEvent_t start, prep_done[2], main_done[2], comb_done[2];
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
                  hipStreamCreateWithFlags(&p.comb, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&p.start, hipEventDisableTiming) == hipSuccess;
        for (int q = 0; q < 2 && ok; q++)
            ok = hipEventCreateWithFlags(&p.prep_done[q], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&p.main_done[q], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&p.comb_done[q], hipEventDisableTiming) == hipSuccess;
        if (!ok) return nullptr;
        made[dev] = true;
    }
    return &sets[dev];
}

extern "C" int ital_score_generic(const ital_gscore_desc* d, hipStream_t stream) {
    if (!d) return ital_fail(-22, "ital_score_generic: null descriptor");
    if (d->n_cand <= 0) return 0;
    const int nr = d->n_picks + 1;
    const int nUmax = d->nE + 1;
    if (d->nE < 0 || nUmax > ITAL_GENERIC_MAX_DIM)
        return ital_fail(-22, "ital_score_generic: orthant dimension exceeds ITAL_GENERIC_MAX_DIM");
    if (nr < 1 || nr > ITAL_GENERIC_MAX_REL)
        return ital_fail(-22, "ital_score_generic: more enumerated variables than ITAL_GENERIC_MAX_REL");
    if (!d->subset_mode && d->nE != d->n_picks)
        return ital_fail(-22, "ital_score_generic: without a change-estimation subset the base set is the batch");
    if (d->fb_mode < 0 || d->fb_mode > 3) return ital_fail(-22, "ital_score_generic: fb_mode must be 0, 1, 2 or 3");
    if (d->fb_mode == 3 && (d->subset_mode || d->mc_rel > 0 || d->mc_fb > 0))
        return ital_fail(-22, "ital_score_generic: the entropy objective enumerates the batch patterns (no subset, no sampling)");
    if ((d->mc_rel > 0 && !d->rel_samples) || (d->mc_fb > 0 && !d->fb_samples))
        return ital_fail(-22, "ital_score_generic: sample lists missing");
    {
        double npat = d->mc_rel > 0 ? (double)d->mc_rel : pow(2.0, nr);
        double nfb = d->fb_mode == 3 ? 0.0 : d->fb_mode == 0 ? 1.0 : (d->mc_fb > 0 ? (double)d->mc_fb : (d->fb_mode == 1 ? pow(2.0, nr) : pow(3.0, nr) - 1));
        if (npat * (2 + nfb) > (double)ITAL_GENERIC_MAX_CALLS)
            return ital_fail(-22, "ital_score_generic: more calls per candidate than ITAL_GENERIC_MAX_CALLS (use the monte-carlo switches)");
    }
    if (!d->jump1 || !d->vk) return ital_fail(-22, "ital_score_generic: stream tables missing");
    GArgs a;
    a.d = *d;
    const int slab = nUmax * (nUmax + 1) / 2 + 2 * nUmax;
    const bool clip = d->clip_cov > 0 && d->clip_cov < 1 && nUmax > 5;
    const int master = clip ? slab + 2 * nUmax + 2 : 0;     // limits, packed correlations, adjacency / order / boundaries
    int stride = slab + fs_doubles(nr) + 16 * (nUmax - 1) + master;
    stride |= 1;
    int chunk = 64;
    while (chunk > 4 && chunk * stride > 1536) chunk >>= 1;   // ~14 KB of LDS per wave (measured best: 16 preparing lanes at t = 4)
    a.chunk = chunk;
    a.stride = stride;
    a.slab = slab;
    a.lat = slab + fs_doubles(nr);
    a.master = clip ? slab + fs_doubles(nr) + 16 * (nUmax - 1) : 0;
    const int fixed = nUmax + nUmax * nUmax + (GN + GR + 1) / 2 + ITAL_GEN_TAILQ;
    a.ldS = nUmax;
    a.wave_doubles = fixed + chunk * stride;
    const size_t lds = (size_t)2 * a.wave_doubles * sizeof(double);
    if (lds > 160 * 1024) return ital_fail(-12, "ital_score_generic: LDS budget exceeded");
    const int64_t blocks = (d->n_cand + 1) / 2;
    // plain mode (no subset): every call of dimension >= 3 has dimension n_picks + 1 -> compile-time evaluator
    const int tfix = (!d->subset_mode && nUmax >= 3 && nUmax <= 6) ? nUmax : 0;
    // ---- three-kernel pipeline: plain mode with a compile-time evaluator, slabs of candidates through the workspace
    {
        double npat = d->mc_rel > 0 ? (double)d->mc_rel : pow(2.0, nr);
        double nfb = d->fb_mode == 3 ? 0.0 : d->fb_mode == 0 ? 1.0 : (d->mc_fb > 0 ? (double)d->mc_fb : (d->fb_mode == 1 ? pow(2.0, nr) : pow(3.0, nr) - 1));
        const int64_t total = (int64_t)(npat * (1 + nfb));
        GPipe g;
        g.total = (int)total;
        g.lat = nUmax * (nUmax + 1) / 2 + nUmax;
        g.R = g.lat + 16 * (nUmax - 1);
        const int64_t per_cand = total * (2 + (int64_t)g.R) + (total + 1) / 2;      // meta, records, list entries
        const int64_t half = d->work_doubles / 2 - 1;                                // two buffers, a counter each
        const int tfix_p = (!d->subset_mode && nUmax >= 3 && nUmax <= ITAL_GEN_TFIX_MAX) ? nUmax : 0;
        if (ITAL_GEN_PIPELINE && tfix_p != 0 && !clip && !d->draw_count && d->work && half >= per_cand) {
            PipeStreams* ps = pipe_streams();
            if (!ps) return ital_fail(-12, "ital_score_generic: cannot create the pipeline's streams");
            int64_t S = half / per_cand;
            if (S > d->n_cand) S = d->n_cand;
            while (S * total > (int64_t)1 << 31) S >>= 1;     // 32-bit list entries
            GArgs ap = a;
            int stride_p = (slab + fs_doubles(nr)) | 1;
            int chunk_p = 64;
            while (chunk_p > 4 && chunk_p * stride_p > 4096) chunk_p >>= 1;   // <= 32 KB of call slabs per wave
            ap.chunk = chunk_p;
            ap.stride = stride_p;
            ap.master = 0;
            ap.wave_doubles = fixed + chunk_p * stride_p;
            const size_t lds_p = (size_t)2 * ap.wave_doubles * sizeof(double);
            const size_t lds_m = (size_t)4 * (g.R + ITAL_GEN_TAILQ) * sizeof(double);
            // a slab is a few hundred candidates: one wave each would leave most of the chip idle during the preparation
            const int npass = (int)((total + chunk_p - 1) / chunk_p);
            g.nsplit = d->mc_fb > 0 ? 1 : (npass < ITAL_GEN_PREP_SPLIT ? (npass < 1 ? 1 : npass) : ITAL_GEN_PREP_SPLIT);
            static ItalLdsFlags prep_flags;
            if (const int rc = ital_raise_lds_limit(reinterpret_cast<const void*>(&gen_prep_kernel), 160 * 1024, prep_flags,
                                                    "ital_score_generic"))
                return rc;
            if (lds_p > 160 * 1024) return ital_fail(-12, "ital_score_generic: LDS budget exceeded");
            if (hipEventRecord(ps->start, stream) != hipSuccess || hipStreamWaitEvent(ps->prep, ps->start, 0) != hipSuccess ||
                hipStreamWaitEvent(ps->main, ps->start, 0) != hipSuccess || hipStreamWaitEvent(ps->comb, ps->start, 0) != hipSuccess)
                return ital_fail(-5, "ital_score_generic: stream synchronisation failed");
            int nslab = 0;
            for (int64_t lo = 0; lo < d->n_cand; lo += S, nslab++) {
                const int buf = nslab & 1;
                double* base = d->work + (size_t)buf * (half + 1);
                g.slab_lo = lo;
                g.slab_n = d->n_cand - lo < S ? d->n_cand - lo : S;
                g.count = reinterpret_cast<unsigned int*>(base);
                g.meta = base + 1;
                g.recs = g.meta + g.slab_n * total * 2;
                g.list = reinterpret_cast<unsigned int*>(g.recs + g.slab_n * total * g.R);
                if (nslab >= 2) (void)hipStreamWaitEvent(ps->prep, ps->comb_done[buf], 0);   // the buffer is free again
                (void)hipMemsetAsync(g.count, 0, 2 * sizeof(unsigned int), ps->prep);
                ITAL_LAUNCH(gen_prep_kernel, dim3((unsigned)((g.slab_n * g.nsplit + 1) / 2)), dim3(128), lds_p, ps->prep, ap, g);
                (void)hipEventRecord(ps->prep_done[buf], ps->prep);
                (void)hipStreamWaitEvent(ps->main, ps->prep_done[buf], 0);
                const unsigned mb = 768;     // 3 workgroups of 4 waves per CU; the waves stride over the slab's list
                const unsigned cap_b = (unsigned)((g.slab_n * total + 3) / 4);     // the whole list, one call per wave
#define ITAL_GEN_MAIN(T_) case T_: ITAL_LAUNCH(gen_main_kernel<T_>, dim3(ITAL_GEN_ONE_TRIP(T_) ? cap_b : mb), dim3(256), lds_m, ps->main, g, d->pair_count); break;
                switch (tfix_p) {
                    ITAL_GEN_MAIN(3) ITAL_GEN_MAIN(4) ITAL_GEN_MAIN(5) ITAL_GEN_MAIN(6) ITAL_GEN_MAIN(7) ITAL_GEN_MAIN(8)
                    ITAL_GEN_MAIN(9) ITAL_GEN_MAIN(10) ITAL_GEN_MAIN(11) ITAL_GEN_MAIN(12) ITAL_GEN_MAIN(13) ITAL_GEN_MAIN(14)
                    ITAL_GEN_MAIN(15) ITAL_GEN_MAIN(16)
                }
#undef ITAL_GEN_MAIN
                ITAL_LAUNCH(gen_main_kernel<0>, dim3(64), dim3(256), lds_m, ps->main, g, d->pair_count);
                (void)hipEventRecord(ps->main_done[buf], ps->main);
                (void)hipStreamWaitEvent(ps->comb, ps->main_done[buf], 0);       // the slab's terms add up under the next slab's sums
                ITAL_LAUNCH(gen_combine_kernel, dim3((unsigned)((g.slab_n + 3) / 4)), dim3(256), 0, ps->comb, ap, g);
                (void)hipEventRecord(ps->comb_done[buf], ps->comb);
                int rc = ital_check_launch("ital_score_generic(pipeline)");
                if (rc) return rc;
            }
            (void)hipStreamWaitEvent(stream, ps->comb_done[0], 0);
            if (nslab >= 2) (void)hipStreamWaitEvent(stream, ps->comb_done[1], 0);
            return 0;
        }
    }
#define ITAL_GEN_LAUNCH(NMAX_, NH_, TFIX_, CLIP_)                                                                             \
    do {                                                                                                               \
        static ItalLdsFlags lds_flags;                                                                                 \
        if (const int rc = ital_raise_lds_limit(reinterpret_cast<const void*>(&score_generic_kernel<NMAX_, NH_, TFIX_, CLIP_>), \
                                                160 * 1024, lds_flags, "ital_score_generic"))                          \
            return rc;                                                                                                 \
        ITAL_LAUNCH((score_generic_kernel<NMAX_, NH_, TFIX_, CLIP_>), dim3((unsigned)blocks), dim3(128), lds, stream, a); \
    } while (0)
    if (clip) {     // grouped probabilities: the instantiations that carry the group passes
        if (nUmax <= 6) ITAL_GEN_LAUNCH(6, 2, 0, true);
        else if (nUmax <= 12) ITAL_GEN_LAUNCH(12, ITAL_GEN_NH12, 0, true);
        else ITAL_GEN_LAUNCH(ITAL_GENERIC_MAX_DIM, 1, 0, true);
    } else {
        switch (tfix) {
            case 3: ITAL_GEN_LAUNCH(6, 2, 3, false); break;
            case 4: ITAL_GEN_LAUNCH(6, 2, 4, false); break;
            case 5: ITAL_GEN_LAUNCH(6, 2, 5, false); break;
            case 6: ITAL_GEN_LAUNCH(6, 2, 6, false); break;
            default:
                if (nUmax <= 6) ITAL_GEN_LAUNCH(6, 2, 0, false);
                else if (nUmax <= 12) ITAL_GEN_LAUNCH(12, ITAL_GEN_NH12, 0, false);
                else ITAL_GEN_LAUNCH(ITAL_GENERIC_MAX_DIM, 1, 0, false);
        }
    }
#undef ITAL_GEN_LAUNCH
    return ital_check_launch("ital_score_generic");
}
