// Device code shared by the general scorer's two translation units (score_generic.hip: the single-kernel form --
// change-estimation subset, clip_cov, more than 16 variables; gen_pipeline.hip: plain mode as a pipeline of kernels):
// the reference's call sequence of a candidate (decode_call), the closed-form simulated update and the standardised
// orthant problem of a call (prepare_call), Genz's COVSRT with its zero-diagonal branch and the grouped limits of
// MVNDFN for a runtime dimension (covsrt_n, group_layout, qmc_eval), the tuning knobs of both units.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

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
// chains per lane and round of the compile-time evaluator for 7 .. 16 variables.
// Rounds 2 - 4 sized these by what 256 / 168 registers held: two to four chains at two or three waves per SIMD (history of
// the measurements: profiles/r3_general_variants.txt).  Round 5 found that most of those registers held nothing the loop
// needs: the scheduler had sunk every chain's running product of interval widths to the end of the round and carried the
// widths of all stages until then (qmc_common.h ITAL_QMC_PIN_FF: 90 of the 244 registers at T = 16).  With the product
// formed per stage the instantiations take 111 (T = 7) .. 170 (T = 16, three chains) registers, and the table below is the
// measured optimum of chains x waves per dimension (20 000 x 64, k = 16, monte_carlo_num_rel = 1, ms per step,
// profiles/r5_variants_wide*.txt): more chains = fuller waves in the Phi^-1 tail branch (121 instructions per pass whatever
// the number of lanes in it: 133 / 122 / 114 vector instructions per pair with 2 / 3 / 4 chains), more waves = more of the
// issue slots filled (0.67 - 0.70 at two waves per SIMD, 0.77 - 0.81 at three, 0.87 at four).
//   T      7     8     9     10    11    12    13    14    15    16
//   r4    17.8  30.0  60.0  107.8 209.1 251.6 315.5 366.7 443.4 509.1   (4 4 3 3 4 4 2 2 3 3 chains at 3 3 3 3 2 2 3 3 2 2 waves)
//   r5    16.5  27.4  53.3   96.1 176.1 211.3 261.5 316.9 373.4 430.6   (4 4 5 5 4 4 4 4 4 3 chains at 4 4 3 3 3 3 3 3 3 3 waves)
#define ITAL_GEN_BIG_NCB(T) ((T) == 9 || (T) == 10 ? 5 : (T) == 16 ? 3 : 4)
#endif
#ifndef ITAL_GEN_BIG_HOTK
#define ITAL_GEN_BIG_HOTK 1
#endif
#ifndef ITAL_GEN_BIG_KEN
// exp coefficients (of ten) held in vector registers by the evaluator above, the rest materialised in place (HotKEn,
// device_math.h): six where the registers are there, two at 13 variables, none at 14 and 15 (four chains at three waves per
// SIMD without scratch; T = 13: 269 -> 261.5 ms with two against none)
#define ITAL_GEN_BIG_KEN(T) ((T) == 13 ? 2 : (T) == 14 || (T) == 15 ? 0 : 6)
#endif
#ifndef ITAL_GEN_BIG_COEF
#define ITAL_GEN_BIG_COEF(T) HotKEn<ITAL_GEN_BIG_KEN(T)>
#endif
#ifndef ITAL_GEN_FIXED_NH
// lattice items per lane and round of the pipeline's evaluator for 3 .. 6 variables: three (six chains) at 5 and 6
// variables, which then run at two waves per SIMD (per step at 40 000 x 512: 13.6 -> 11.3 ms, 26.5 -> 20.5 ms; four items
// lose again); two at 3 and 4, three waves per SIMD
// Round 3 (registers freed by lit_s): three items = six chains at every dimension, see ITAL_GEN_MAIN_WAVES
// Round 5: two items (four chains) at six variables, four waves per SIMD: as qmc_main_kernel<6> (score.hip ITAL_QMC_MAIN_NH)
#define ITAL_GEN_FIXED_NH(T) ((T) == 6 ? 2 : 3)
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
// waves per SIMD the lattice-sum kernels aim at: four up to 4 variables (120 registers with six chains), three at 5 and 6;
// 6 .. 8 variables four again since round 5 (110 - 126 registers), three beyond (see ITAL_GEN_BIG_NCB)
#define ITAL_GEN_MAIN_WAVES(T) ((T) > 0 && (T) < 5 ? 4 : ((T) >= 6 && (T) <= 8 ? 4 : 3))
#endif
#ifndef ITAL_GEN_TFIX_MAX
#define ITAL_GEN_TFIX_MAX 16   // largest dimension the pipeline takes (plain mode)
#endif
#ifndef ITAL_GEN_PREP_SPLIT
#define ITAL_GEN_PREP_SPLIT 4   // preparation waves per candidate in the pipeline
#endif
#ifndef ITAL_GEN_PIPE_SUBSET
#define ITAL_GEN_PIPE_SUBSET 1   // change-estimation subsets (without clip_cov) through the pipeline's wide form
#endif
#ifndef ITAL_GEN_SUB_MAX
#define ITAL_GEN_SUB_MAX 13      // ... up to this many variables (subset + batch + candidate; the instantiations with MVNPHI's
                                 // far-tail branch fit their register budget without scratch up to here)
#endif
#ifndef ITAL_GEN_PREP_PU
#define ITAL_GEN_PREP_PU 1   // wide form, perfect user, <= 16 patterns per candidate: the cooperative preparation (gen_prep_pu_kernel)
#endif
#ifndef ITAL_GEN_PIPELINE
#define ITAL_GEN_PIPELINE 1   // plain mode, 3 .. 16 variables: prepare / lattice sums / combine as three kernels on streams of their own
#endif
#ifndef ITAL_GEN_COUNT_SYNC
#define ITAL_GEN_COUNT_SYNC 1   // fast form: read the length of list U back after a slab's verdicts and launch only the chunks
#endif                          // that hold entries (0: the worst-case number of chunk iterations, no host wait -- round 2 .. 5)
#ifndef ITAL_GEN_EARLY
#define ITAL_GEN_EARLY 1   // decide saturated calls from the standardised limits, before COVSRT
#endif

#ifndef ITAL_GEN_FN
// the larger helpers (COVSRT, the group layout, the saturation test): out of line by default -- the single kernel calls them
// from several places; gen_pipeline.hip, whose kernels call each of them once, inlines them (an out-of-line callee saves
// registers on a stack: scratch memory)
#define ITAL_GEN_FN static __device__
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
    int yl;            // single kernel: LDS doubles per wave for the chains' conditioned values (qmc_eval_lds / qmc_exact_lds)
};

// ------------------------------------------------------------------------------------------------ COVSRT, runtime n
ITAL_GEN_FN void rcswp_n(int n, int p, int q, double* cov, double* lim, unsigned& infi) {
    double tmp = lim[p]; lim[p] = lim[q]; lim[q] = tmp;
    unsigned bp = (infi >> p) & 1u, bq = (infi >> q) & 1u;
    infi = (infi & ~((1u << p) | (1u << q))) | (bq << p) | (bp << q);
    tmp = cov[pidx(p, p)]; cov[pidx(p, p)] = cov[pidx(q, q)]; cov[pidx(q, q)] = tmp;
    for (int j = 0; j < p; j++) { tmp = cov[pidx(p, j)]; cov[pidx(p, j)] = cov[pidx(q, j)]; cov[pidx(q, j)] = tmp; }
    for (int i = p + 1; i < q; i++) { tmp = cov[pidx(i, p)]; cov[pidx(i, p)] = cov[pidx(q, i)]; cov[pidx(q, i)] = tmp; }
    for (int i = q + 1; i < n; i++) { tmp = cov[pidx(i, p)]; cov[pidx(i, p)] = cov[pidx(i, q)]; cov[pidx(i, q)] = tmp; }
}

// (the limit types travel by value -- in, and the re-ordered ones out: a reference into the caller's registers would
// force them into scratch memory across this out-of-line call)
ITAL_GEN_FN unsigned covsrt_n(int n, double* cov, double* lim, double* y, unsigned infi) {
    const double SQTWPI = 2.506628274631001, EPS = 1e-10;
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
    return infi;
}

// After COVSRT: which rows close a group of MVNDFN (a row whose successor has a positive diagonal, or the last row),
// and the factor re-packed so that column (closing row of group g) carries the coefficient of group g -- the
// evaluator can then index its y registers by row.  Bit i of the result = row i closes a group.
ITAL_GEN_FN unsigned group_layout(int n, double* cov) {
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
ITAL_GEN_FN CallInfo decode_call(const ital_gscore_desc& d, int64_t p, int call, int cpp, int npre, int nr, int npat) {
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
ITAL_GEN_FN int early_decision(int n, const double* lim, unsigned infi) {
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
// Integrand identically 1 / 0 after COVSRT?  Every conditional limit stays beyond +-37 for any |y| <= 9.  Returns 2 / 4 / 0.
ITAL_GEN_FN int saturation_n(int n, const double* cov, const double* lim, unsigned infi) {
    bool sat1 = true, sat0 = false;
    for (int i = 0; i < n; i++) {
        double bound = 0;
        for (int j = 0; j < i; j++) bound += fabs(cov[pidx(i, j)]) * 9.0;
        const bool lower = (infi >> i) & 1u;
        if (lower) {
            if (!(lim[i] + bound < -37.0)) sat1 = false;
            if (lim[i] - bound > 37.0) sat0 = true;
        } else {
            if (!(lim[i] - bound > 37.0)) sat1 = false;
            if (lim[i] + bound < -37.0) sat0 = true;
        }
    }
    return sat0 ? 4 : (sat1 ? 2 : 0);
}

__device__ __forceinline__ void finish_call(int n, double* cov, double* lim, double* y, Prep& out) {
    if (n == 2) {
        out.value = bvn_orthant(lim[0], lim[1], out.infi & 1u, (out.infi >> 1) & 1u, cov[pidx(1, 0)]);
        out.flags = 1;
        return;
    }
    out.infi = covsrt_n(n, cov, lim, y, out.infi);
    out.closes = group_layout(n, cov);
    const int sat = saturation_n(n, cov, lim, out.infi);
    if (sat) out.flags = sat;
}

// clip_cov (reference ital/ital.py:386-429, :590-616): connected components of |corr| > clip over the n variables, in
// group_cov's order -- seeds ascending, members in breadth-first layers, each layer ascending.  Writes the member order
// and the group boundaries; returns the number of groups.
static __device__ int clip_groups(int n, const double* cor, double clip, int* adj, int* gorder, int* gstart) {
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
static __device__ ITAL_GEN_NOINLINE Prep build_group(int g, const double* mlim, const double* mcor, unsigned infi_full, const int* gorder,
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

// The three options of the descriptor prepare_call reads, by value: a reference to the descriptor into a function that is not
// inlined made the compiler copy the kernel's whole argument block (408 B) to scratch memory in the instantiations that use
// all three (round 5).
struct PrepOpts {
    double noise, clip_cov;
    int subset_mode;
};
__device__ __forceinline__ PrepOpts prep_opts(const ital_gscore_desc& d) { return PrepOpts{d.noise, d.clip_cov, d.subset_mode}; }

// Prepares one call in the lane's slab: cov (packed, n(n+1)/2), lim (n), y (n); scratch fs for the update.
template <bool CLIP>
static __device__ ITAL_GEN_NOINLINE Prep prepare_call(const PrepOpts d, const CallInfo ci, int nU, int nr, int ldS, const double* muU,
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
static __device__ double qmc_eval(int n, const double* __restrict__ slab, unsigned infi, unsigned closes,
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

// The same pass with the conditioned values of the chains in LDS (yl: [2][GN - 1][64] doubles of wave-private memory) and
// plain loops over the runtime dimension: one lattice item per lane and round (a point and its antithetic partner), the
// same operations in the same order as qmc_eval<NMAX, 1>.  For the rare calls with linearly dependent variables
// (gen_pipeline.hip): no unrolled copies of the stage for 20 dimensions, no register arrays that end up in scratch memory.
static __device__ double qmc_eval_lds(int n, const double* __restrict__ slab, unsigned infi, unsigned closes,
                                      const double* __restrict__ lat, int lane, double* __restrict__ tailq, double* __restrict__ yl) {
    const int ndim = n - 1;
    const int prime = P_TAB[(ndim < 10 ? ndim : 10) - 1];
    const double* cf = slab;
    const double* lm = slab + n * (n + 1) / 2;
    const int items = 8 * prime;
    double* y0 = yl + lane;
    double* y1 = yl + (GN - 1) * 64 + lane;
    double acc = 0.0;
    for (int base = 0; base < items; base += 64) {
        const int item = base + lane;
        const bool ok = item < items;
        const int it = ok ? item : 0;
        const int sft = it / prime;
        const int kk = it - sft * prime + 1;
        const int so = sft * ndim;
        double ff[2] = {1.0, 1.0}, ai[2] = {0.0, 0.0}, bi[2] = {0.0, 0.0};
        bool dead[2] = {!ok, !ok};
        bool infa = false, infb = false;   // wave-uniform: the open group has a lower / an upper limit (MVNDFN)
        int ik = 0;                        // groups closed so far = lattice coordinate of the open group
        for (int i = 0; i < n; i++) {
            const bool lower = (infi >> i) & 1u;
            const bool close = (closes >> i) & 1u;
            const bool last = i == n - 1;
            double sc0 = 0, sc1 = 0;
            for (int j = 0; j < i; j++) {
                const double c = cf[pidx(i, j)];
                sc0 = fma(c, y0[j * 64], sc0);
                sc1 = fma(c, y1[j * 64], sc1);
            }
            const double z0 = lm[i] - sc0, z1 = lm[i] - sc1;
            if (lower) { ai[0] = infa ? fmax(ai[0], z0) : z0; ai[1] = infa ? fmax(ai[1], z1) : z1; infa = true; }
            else { bi[0] = infb ? fmin(bi[0], z0) : z0; bi[1] = infb ? fmin(bi[1], z1) : z1; infb = true; }
            if (close) {
                double xh = 0;
                if (!last) {
                    const double v = kk * lat[so + ik] + lat[8 * ndim + so + ik];
                    const double fr = v - floor(v);
                    xh = fabs(2 * fr - 1);
                }
                double pin[2];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const double dd = infa ? mvn_phi(ai[c]) : 0.0;
                    const double ee = infb ? mvn_phi(bi[c]) : 1.0;
                    const double w = ee - dd;
                    dead[c] = dead[c] || !(w > 0);
                    ff[c] *= w;
                    const double x = (c & 1) ? 1 - xh : xh;
                    pin[c] = fma(x, w, dd);
                }
                if (!last) {
                    double outv[2];
                    phinv_wave<2>(pin, outv, tailq, lane);
                    y0[i * 64] = outv[0];
                    y1[i * 64] = outv[1];
                }
                infa = false; infb = false;
                ik++;
            } else if (!last) {
                y0[i * 64] = 0.0;
                y1[i * 64] = 0.0;
            }
        }
        acc += dead[0] ? 0.0 : ff[0];
        acc += dead[1] ? 0.0 : ff[1];
    }
    return wave_sum(acc) / (16.0 * prime);
}

// One MVNDST pass for a call of compile-time dimension T whose rows all close their own group (no linearly dependent
// variable): the evaluator of the perfect-user fast path (score.hip) -- factor and limits as wave-uniform scalars, fully
// unrolled, NHF lattice items x antithetic partner per lane.
template <int T, int NHF = 2, bool FL = false, bool CF = false>
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
    typename std::conditional<CF, WithCF<HotK>, HotK>::type kk;      // CF: MVNPHI's far-tail branch (device_math.h WithCF)
    kk.load();
    const double acc = qmc_lane_sum<T, decltype(kk), NHF, FL>(lat, cf, lm, infi, tailq, lane, kk);
#else
    const double acc = qmc_lane_sum<T, LitK, NHF, FL>(lat, cf, lm, infi, tailq, lane);
#endif
    return wave_sum(acc) / (16.0 * PRIME);
}

template <int T>
static __device__ double qmc_eval_fixed(const double* __restrict__ slab, unsigned infi, const double* __restrict__ lat, int lane,
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
// (State and tables by value: a reference would keep the caller's generator state in scratch memory.)
static __device__ ITAL_GEN_NOINLINE void make_lattice(const long long* jump1, const double* vk, MrgState sti, unsigned before, int n, double* L) {
    for (int bit = 0; before != 0; bit++, before >>= 1)
        if (before & 1u) mrg_apply(sti, jump1 + bit * 18);
    MrgStateF st = mrg_to_f(sti);
    const int ndim = n - 1;
    for (int j = 0; j < ndim; j++) L[j] = vk[n * GN + j];
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

}  // namespace ital
