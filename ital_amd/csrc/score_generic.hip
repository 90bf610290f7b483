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
#include "gen_common.h"
#include "qmc_exact.h"

namespace ital {

#ifndef ITAL_GEN_WAVES
#define ITAL_GEN_WAVES 2   // waves per SIMD the single kernel aims at (the runtime-dimension evaluator spills at three)
#endif
// (Until round 3 this kernel also had instantiations with a compile-time evaluator for plain mode with 3 .. 6 variables;
// that case is gen_pipeline.hip's -- without a workspace it runs here on the runtime-dimension evaluator.)
template <int NMAX, int NH, bool CLIP>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(ITAL_GEN_WAVES, ITAL_GEN_WAVES))) void score_generic_kernel(GArgs a) {
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
    // more than 6 variables: the chains' conditioned values live in LDS (qmc_eval_lds: no 20-fold unrolled stage, no
    // register arrays that end up in scratch memory)
    // (up to 6 variables that area exists only with label_estimation 'optimistic' / 'pessimistic': a.yl doubles, for the
    // recomputation below)
    double* yl = tailq + ITAL_GEN_TAILQ;
    double* slabs = yl + a.yl;
    // label_estimation 'optimistic' / 'pessimistic' (plain mode; the reference ignores them with a subset, ital.py:227-275)
    // compare terms for EXACT equality (ital.py:210-215): an orthant probability whose flat sum lands within 1e-9 of 0 or 1 is
    // formed again in MVKBRV's own serial order (qmc_exact.h) -- also the probability of a group of clip_cov, whose product
    // with the other groups' is the call's value.  Round 6: until then this kernel took the flat sums.
    const bool exact = d.subset_mode == 0 && d.label_mode != 0 && d.fb_mode != 3;
    const int ystride = NMAX > 6 ? GN - 1 : NMAX - 1;
    auto lattice_sum = [&](int n_c, const double* slab_c, unsigned infi_c, unsigned closes_c) -> double {
        double v;
        if constexpr (NMAX > 6) v = qmc_eval_lds(n_c, slab_c, infi_c, closes_c, slab_c + a.lat, lane, tailq, yl);
        else v = qmc_eval<NMAX, NH>(n_c, slab_c, infi_c, closes_c, slab_c + a.lat, lane, tailq);
        if (exact && (v > 1.0 - EXACT_BAND || v < EXACT_BAND))       // (wave-uniform)
            v = qmc_exact_lds(n_c, slab_c, infi_c, closes_c, slab_c + a.lat, lane, tailq, yl, ystride, tailq + 256);
        return v;
    };

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
            else pp = prepare_call<CLIP>(prep_opts(d), ci, nU, nr, ldS, muU, SigU, usort, ipos, clamp_prior, slab, slab + a.slab,
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
                make_lattice(d.jump1, d.vk, rng, call_base, pp.n, slabs + (size_t)lane * a.stride + a.lat);
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
                            if (!(gp.flags & 7)) make_lattice(d.jump1, d.vk, rng, call_base + gdone, gp.n, slab + a.lat);
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
                        const double v = lattice_sum(n_c, slab_c, infi_c, closes_c);
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
                value = lattice_sum(n_c, slab_c, infi_c, closes_c);
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

}  // namespace ital

using namespace ital;

static int fs_doubles(int nr) { return nr * nr + 2 * nr; }

// gen_pipeline.hip: plain mode with 3 .. 16 variables as a pipeline of kernels through the workspace.  Returns 1 when the
// step is not the pipeline's (subset, clip_cov, counting pass, no or too small a workspace): the single kernel below runs.
int ital_gen_pipeline(const ital_gscore_desc* d, hipStream_t stream);

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
    // the jump-ahead table has ITAL_JUMP_BITS rows (one 3x3 matrix pair per bit of the offset): a candidate's offset is at most
    // (its list position) x draws_out.  With explicit list positions (gpos) / offsets (draw_off) the caller vouches for them.
    if (!d->gpos && !d->draw_off && d->draws_out > 0) {
        const long double most = (long double)(d->pos_offset + d->n_cand) * (long double)d->draws_out;
        if (most >= (long double)((uint64_t)1 << ITAL_JUMP_BITS))
            return ital_fail(-22, "ital_score_generic: stream offset beyond 2^ITAL_JUMP_BITS uniforms (list position x draws per candidate)");
    }
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
    // conditioned values of the chains in LDS: above 6 variables always (qmc_eval_lds); below only for the recomputation of
    // label_estimation 'optimistic' / 'pessimistic' (qmc_exact_lds)
    const bool exact = d->subset_mode == 0 && d->label_mode != 0 && d->fb_mode != 3;
    a.yl = nUmax > 6 ? 2 * (GN - 1) * 64 : (exact ? 2 * 5 * 64 : 0);
    const int fixed = nUmax + nUmax * nUmax + (GN + GR + 1) / 2 + ITAL_GEN_TAILQ + a.yl;
    a.ldS = nUmax;
    a.wave_doubles = fixed + chunk * stride;
    const size_t lds = (size_t)2 * a.wave_doubles * sizeof(double);
    if (lds > 160 * 1024) return ital_fail(-12, "ital_score_generic: LDS budget exceeded");
    const int64_t blocks = (d->n_cand + 1) / 2;
    // ---- plain mode with a compile-time evaluator: the pipeline of kernels (gen_pipeline.hip)
    if (ITAL_GEN_PIPELINE && !clip && !d->draw_count) {
        const int rc = ital_gen_pipeline(d, stream);
        if (rc <= 0) return rc;
    }
#define ITAL_GEN_LAUNCH(NMAX_, NH_, CLIP_)                                                                             \
    do {                                                                                                               \
        static ItalLdsFlags lds_flags;                                                                                 \
        if (const int rc = ital_raise_lds_limit(reinterpret_cast<const void*>(&score_generic_kernel<NMAX_, NH_, CLIP_>), \
                                                160 * 1024, lds_flags, "ital_score_generic"))                          \
            return rc;                                                                                                 \
        ITAL_LAUNCH((score_generic_kernel<NMAX_, NH_, CLIP_>), dim3((unsigned)blocks), dim3(128), lds, stream, a); \
    } while (0)
    if (clip) {     // grouped probabilities: the instantiations that carry the group passes
        if (nUmax <= 6) ITAL_GEN_LAUNCH(6, 2, true);
        else ITAL_GEN_LAUNCH(ITAL_GENERIC_MAX_DIM, 1, true);
    } else {
        // up to 6 variables: two lattice items per lane and round, the chains' conditioned values in registers; beyond: one
        // item, the values in LDS (one instantiation for 7 .. 20 variables: the loops run over the runtime dimension)
        if (nUmax <= 6) ITAL_GEN_LAUNCH(6, 2, false);
        else ITAL_GEN_LAUNCH(ITAL_GENERIC_MAX_DIM, 1, false);
    }
#undef ITAL_GEN_LAUNCH
    return ital_check_launch("ital_score_generic");
}
