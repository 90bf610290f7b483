// Pieces shared by the lattice-rule scorers (score.hip: perfect-user fast path; score_generic.hip: every other
// user model / the change-estimation subset): Genz's lattice sizes, packed-triangle indexing, wave-uniform values and
// the wave-compacted Phi^-1.
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.h"

#ifndef ITAL_QMC_NH
#define ITAL_QMC_NH 2      // lattice items per lane and round (each with its antithetic partner)
#endif
#ifndef ITAL_QMC_TRIM_LAST
#define ITAL_QMC_TRIM_LAST 1   // last round of a call's lattice points with only as many chains per lane as it needs
#endif

#ifndef ITAL_QMC_PIN_FF
// The running product of a chain's interval widths is formed at every stage (an empty asm ties the value down).  Left to
// itself the scheduler sinks the T multiplications of every chain to the end of the round and carries all the widths until
// then: 2 registers per chain and stage (found in round 5: 90 of the 244 registers of the T = 16 lattice sums).
#define ITAL_QMC_PIN_FF 1
#endif
#ifndef ITAL_QMC_FLIP
// Every variable of an orthant call as an UPPER-bounded one (the compile-time lattice sums: perfect-user scorer, regular
// calls of the general scorer): a variable bounded below enters negated -- its limit, its row and its column of the factor
// change sign (in the call's record, or when the wave unpacks it), its lattice shifts move by 1/2, i.e. the point and its
// antithetic partner swap that coordinate.  Phi^-1(d + x (1 - d)) = -Phi^-1((1 - x) Phi(-z)): the same sum, and the
// lattice loop loses the per-stage selects between the two forms (5 of ~41 vector instructions per chain and stage,
// measured 3 % of the t = 4 kernel); the interval width of such a variable is Phi(-z) where MVNDFN forms 1 - Phi(z).
#define ITAL_QMC_FLIP 1
#endif

namespace ital {

constexpr int P_TAB[10] = {31, 47, 73, 113, 173, 263, 397, 593, 907, 1361};

__device__ __forceinline__ int pidx(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower, 0-based, j <= i

__device__ __forceinline__ double uniform_f64(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Width of the interval of a variable that entered NEGATED (ITAL_QMC_FLIP): MVNDFN forms it as 1 - Phi(z) from the rounded
// Phi(z) = 1 - p (z > 0) -- exactly 0 once p <= 2^-54, a multiple of 2^-53 below ~1e-3 -- and the reference's logic hangs on
// such exact zeros (`mi == 0` in the pessimistic label estimation, log(0 + eps)).  The negated form computes Phi(-z) = p
// itself; 1 - (1 - w) reproduces MVNDFN's value bit for bit (w >= 1/2: both subtractions are exact, nothing changes).
// `flipped` is wave-uniform: a scalar branch around two additions per chain (the empty asm keeps it a branch).
template <int NCB>
__device__ __forceinline__ void flip_width(double (&w)[NCB], bool flipped) {
    if (flipped) {
        __asm__ volatile("");
#pragma unroll
        for (int c = 0; c < NCB; c++) w[c] = 1.0 - (1.0 - w[c]);
    }
}

// Four independent Phi^-1 arguments per lane.  Every lane runs the cheap central branch of AS241; the ~15 % of the
// arguments that fall into the tails (|p - 1/2| > 0.425) are compacted across the wave through LDS so that the
// expensive log/sqrt branch runs on full waves of tail arguments only (typically once per 256 inversions instead of
// once per 64).
// q: wave-private LDS queue of 64*NC doubles.
template <int NC, class K, int GRP = 0>
__device__ __forceinline__ void phinv_wave(const double (&p)[NC], double (&out)[NC], double* __restrict__ q, int lane,
                                           const K& kk) {
    bool need[NC];
    int slot[NC];
    int total = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        // GRP > 0: the chains are evaluated in groups of GRP -- the scheduler may interleave the rationals of a group, not of
        // all NC chains (whose temporaries would all be live at once: eval_chains_big)
        if (GRP > 0 && c > 0 && c % GRP == 0) __builtin_amdgcn_sched_barrier(0);
        const double qc = p[c] - 0.5;
        need[c] = !(fabs(qc) <= 0.425);
        out[c] = phinv_central_q(qc);   // garbage (never a trap) for tail arguments: replaced below
        const unsigned long long m = __ballot(need[c]);
        slot[c] = total + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        total += __popcll(m);
        if (need[c]) q[slot[c]] = p[c];
    }
    if (total == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int s0 = 0; s0 < total; s0 += 64) {
        const int sl = s0 + lane;
        if (sl < total) q[sl] = phinv_tail(q[sl], kk);   // in place: each lane rewrites the slot it read
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int c = 0; c < NC; c++)
        if (need[c]) out[c] = q[slot[c]];
}

template <int NC>
__device__ __forceinline__ void phinv_wave(const double (&p)[NC], double (&out)[NC], double* __restrict__ q, int lane) {
    phinv_wave<NC>(p, out, q, lane, LitK());
}

// The integrand of NCB lattice points per lane (MVNDFN for one-sided limits): sequential conditioning over the T
// variables, every chain independent of the others; returns the lane's sum of the integrand values.  cf: packed strict
// lower triangle of the row-scaled factor, lm: scaled limits (wave-uniform), bit i of infi: variable i is bounded below.
// FL: the call is in the all-upper form (ITAL_QMC_FLIP); bit i of infi then marks the variables that entered negated.
template <int T, int NCB, class K, bool FL = false>
__device__ __forceinline__ double eval_chains(const double (&xx)[NCB][(T - 1 > 0 ? T - 1 : 1)], bool (&dead)[NCB],
                                              const double (&cf)[(T * (T - 1) / 2 > 0 ? T * (T - 1) / 2 : 1)],
                                              const double (&lm)[T], unsigned infi_c, double* tailq, int lane, const K& kk) {
    // a chain whose interval closes (w == 0, MVNDFN's early return) keeps ff == 0 through the finite values that follow
    // (Phi^-1 of exactly 0 or 1 is -+9 here); chains beyond the end of the lattice start from 0
    double yy[NCB][(T - 1 > 0 ? T - 1 : 1)], ff[NCB];
#pragma unroll
    for (int c = 0; c < NCB; c++) ff[c] = dead[c] ? 0.0 : 1.0;
#pragma unroll
    for (int i = 0; i < T; i++) {
        const bool lower = (infi_c >> i) & 1u;
        double pin[NCB], ph[NCB];
#pragma unroll
        for (int c = 0; c < NCB; c++) {
            double sc = 0;
#pragma unroll
            for (int j = 0; j < i; j++) sc = fma(cf[i * (i - 1) / 2 + j], yy[c][j], sc);
            ph[c] = mvn_phi_lat(lm[i] - sc, kk);
        }
        if (FL) flip_width<NCB>(ph, lower);
#pragma unroll
        for (int c = 0; c < NCB; c++) {
            const double d = (!FL && lower) ? ph[c] : 0.0;
            const double w = (!FL && lower) ? 1.0 - ph[c] : ph[c];
            ff[c] *= w;
            if (ITAL_QMC_PIN_FF) __asm__ volatile("" : "+v"(ff[c]));    // (the product is formed now: see eval_chains_big)
            if (i < T - 1) pin[c] = FL ? xx[c][i] * w : fma(xx[c][i], w, d);   // a dead chain (w == 0) just inverts d: finite, discarded
        }
        if (i < T - 1) {
            double out[NCB];
            phinv_wave<NCB>(pin, out, tailq, lane, kk);
#pragma unroll
            for (int c = 0; c < NCB; c++) yy[c][i] = out[c];
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < NCB; c++) acc += ff[c];
    return acc;
}

// Sum over this lane's share of the 16 P lattice points of one orthant call of compile-time dimension T (8 randomly
// shifted Korobov lattices of P points, each point with its antithetic partner).  lat: [8][NDIM] permuted generators, then
// [8][NDIM] shifts; cf / lm / infi as eval_chains.  The caller adds the lanes up and divides by 16 P.
template <int T, class K = LitK, int NH_ = ITAL_QMC_NH, bool FL = false>
__device__ __forceinline__ double qmc_lane_sum(const double* __restrict__ lat,
                                               const double (&cf)[(T * (T - 1) / 2 > 0 ? T * (T - 1) / 2 : 1)],
                                               const double (&lm)[T], unsigned infi_c, double* __restrict__ tailq, int lane,
                                               const K& kk = K()) {
    constexpr int NDIM = T - 1, PRIME = P_TAB[(NDIM < 10 ? NDIM : 10) - 1];
    double acc = 0.0;
    // NC = 2*NH independent chains per lane: NH lattice items, each with its antithetic partner.  The 16 P
    // chains of a call rarely fill whole rounds of 64 NC chains (t = 4: 1168 = 4.56 x 256): the last round runs
    // with just the chains it needs (3 per lane instead of 4 at t = 4: 19 chain slots per lane instead of 20),
    // whole items first, the two chains of the left-over items on neighbouring lanes.
    constexpr int NH = NH_, NC = 2 * NH;
    constexpr int NITEM = 8 * PRIME, FULL = (2 * NITEM) / (64 * NC), REST = 2 * NITEM - FULL * 64 * NC;
    constexpr int NCL = ITAL_QMC_TRIM_LAST ? (REST + 63) / 64 : (REST > 0 ? NC : 0);
    // (a last round that needs all NC chains anyway is simply one more trip of this loop: one copy of the code)
    constexpr int LOOP_END = (NCL == NC ? FULL + 1 : FULL) * 64 * NH;
    for (int base = 0; base < LOOP_END; base += 64 * NH) {
        double xx[NC][NDIM];
        bool dead[NC];
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int item = base + 64 * h + lane;
            const bool ok = NCL != NC || item < NITEM;
            const int it = ok ? item : 0;
            const int sft = it / PRIME;
            const int k = it - sft * PRIME + 1;
#pragma unroll
            for (int j = 0; j < NDIM; j++) {
                const double v = k * lat[sft * NDIM + j] + lat[8 * NDIM + sft * NDIM + j];
                const double fr = v - floor(v);
                xx[2 * h][j] = fabs(2 * fr - 1);
                xx[2 * h + 1][j] = 1 - xx[2 * h][j];
            }
            dead[2 * h] = dead[2 * h + 1] = !ok;
        }
        acc += eval_chains<T, NC, K, FL>(xx, dead, cf, lm, infi_c, tailq, lane, kk);
    }
    if (NCL > 0 && NCL != NC) {
        constexpr int NCLA = NCL > 0 ? NCL : 1;
        double xx[NCLA][NDIM];
        bool dead[NCLA];
        constexpr int base = FULL * 64 * NH;
#pragma unroll
        for (int h = 0; h < NCL / 2; h++) {              // whole items: both antithetic chains on this lane
            const int item = base + 64 * h + lane;
            const bool ok = item < NITEM;
            const int it = ok ? item : 0;
            const int sft = it / PRIME;
            const int k = it - sft * PRIME + 1;
#pragma unroll
            for (int j = 0; j < NDIM; j++) {
                const double v = k * lat[sft * NDIM + j] + lat[8 * NDIM + sft * NDIM + j];
                const double fr = v - floor(v);
                xx[2 * h][j] = fabs(2 * fr - 1);
                xx[2 * h + 1][j] = 1 - xx[2 * h][j];
            }
            dead[2 * h] = dead[2 * h + 1] = !ok;
        }
        if (NCL & 1) {                                    // left-over items: one chain each on lanes 2i, 2i + 1
            const int item = base + 64 * (NCL / 2) + (lane >> 1);
            const bool ok = item < NITEM;
            const int it = ok ? item : 0;
            const int sft = it / PRIME;
            const int k = it - sft * PRIME + 1;
#pragma unroll
            for (int j = 0; j < NDIM; j++) {
                const double v = k * lat[sft * NDIM + j] + lat[8 * NDIM + sft * NDIM + j];
                const double fr = v - floor(v);
                const double x = fabs(2 * fr - 1);
                xx[NCLA - 1][j] = (lane & 1) ? 1 - x : x;
            }
            dead[NCLA - 1] = !ok;
        }
        acc += eval_chains<T, NCLA, K, FL>(xx, dead, cf, lm, infi_c, tailq, lane, kk);
    }
    return acc;
}

// The lattice sum for the larger batch dimensions of the perfect-user kernel (T = 7, 8), where the form above runs out of
// registers (six chains: 6 (T-1) lattice coordinates + 6 (T-1) conditioned values = 168 VGPRs at T = 8 before any
// arithmetic; it spilled 40 VGPRs / 62 SGPRs at two waves per SIMD).  Here a lane keeps per lattice item only the point
// number and the offset of its shift; the coordinate of stage i is formed at stage i from the lattice in LDS (two
// broadcast reads per item: 64 consecutive items span at most two shifts) and serves the point and its antithetic partner.
// CFL: the factor and the limits are read from LDS at use as well (`slab`: NCOR factor values, then T limits) instead
// of living in 2 (NCOR + T) scalar registers.
template <int T, int NI, bool CFL, class K, bool FL = false>
__device__ __forceinline__ double eval_items_ps(const int (&kq)[NI], const int (&so)[NI], const bool (&ok)[NI],
                                                const double* __restrict__ lat,
                                                const double (&cf)[(T * (T - 1) / 2 > 0 ? T * (T - 1) / 2 : 1)],
                                                const double (&lm)[T], const double* __restrict__ slab, unsigned infi_c,
                                                double* tailq, int lane, const K& kk) {
    constexpr int NDIM = T - 1, NCB = 2 * NI, NCOR = T * (T - 1) / 2;
    double yy[NCB][NDIM], ff[NCB];
#pragma unroll
    for (int c = 0; c < NCB; c++) ff[c] = ok[c >> 1] ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < T; i++) {
        const bool lower = (infi_c >> i) & 1u;
        if (CFL) __asm__ volatile("" ::: "memory");   // keeps this row's factor loads inside the lattice loop
        const double lmi = CFL ? slab[NCOR + i] : lm[i];
        double sc[NCB];
#pragma unroll
        for (int c = 0; c < NCB; c++) sc[c] = 0;
#pragma unroll
        for (int j = 0; j < i; j++) {
            const double cij = CFL ? slab[i * (i - 1) / 2 + j] : cf[i * (i - 1) / 2 + j];
#pragma unroll
            for (int c = 0; c < NCB; c++) sc[c] = fma(cij, yy[c][j], sc[c]);
        }
        double pin[NCB], ph[NCB];
#pragma unroll
        for (int c = 0; c < NCB; c++) ph[c] = mvn_phi_lat(lmi - sc[c], kk);
        if (FL) flip_width<NCB>(ph, lower);
#pragma unroll
        for (int h = 0; h < NI; h++) {
            double x0 = 0;
            if (i < T - 1) {
                const double v = kq[h] * lat[so[h] + i] + lat[8 * NDIM + so[h] + i];
                const double fr = v - floor(v);
                x0 = fabs(2 * fr - 1);
            }
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const int c = 2 * h + a;
                const double d = (!FL && lower) ? ph[c] : 0.0;
                const double w = (!FL && lower) ? 1.0 - ph[c] : ph[c];
                ff[c] *= w;
                if (ITAL_QMC_PIN_FF) __asm__ volatile("" : "+v"(ff[c]));
                if (i < T - 1) pin[c] = FL ? (a ? 1 - x0 : x0) * w : fma(a ? 1 - x0 : x0, w, d);
            }
        }
        if (i < T - 1) {
            double out[NCB];
            phinv_wave<NCB>(pin, out, tailq, lane, kk);
#pragma unroll
            for (int c = 0; c < NCB; c++) yy[c][i] = out[c];
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < NCB; c++) acc += ff[c];
    return acc;
}

template <int T, class K, int NH, bool CFL, bool FL = false>
__device__ __forceinline__ double qmc_lane_sum_ps(const double* __restrict__ lat,
                                                  const double (&cf)[(T * (T - 1) / 2 > 0 ? T * (T - 1) / 2 : 1)],
                                                  const double (&lm)[T], const double* __restrict__ slab, unsigned infi_c,
                                                  double* __restrict__ tailq, int lane, const K& kk) {
    constexpr int NDIM = T - 1, PRIME = P_TAB[(NDIM < 10 ? NDIM : 10) - 1];
    constexpr int NITEM = 8 * PRIME, FULL = NITEM / (64 * NH), REST = NITEM - FULL * 64 * NH;
    constexpr int NIL = (REST + 63) / 64;              // items per lane of the last, partial round
    constexpr int LOOP_END = (NIL == NH ? FULL + 1 : FULL) * 64 * NH;
    double acc = 0.0;
    for (int base = 0; base < LOOP_END; base += 64 * NH) {
        int kq[NH], so[NH];
        bool ok[NH];
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int item = base + 64 * h + lane;
            ok[h] = NIL != NH || item < NITEM;
            const int it = ok[h] ? item : 0;
            const int sft = it / PRIME;
            kq[h] = it - sft * PRIME + 1;
            so[h] = sft * NDIM;
        }
        acc += eval_items_ps<T, NH, CFL, K, FL>(kq, so, ok, lat, cf, lm, slab, infi_c, tailq, lane, kk);
    }
    if (NIL > 0 && NIL != NH) {
        constexpr int NILA = NIL > 0 ? NIL : 1;
        int kq[NILA], so[NILA];
        bool ok[NILA];
#pragma unroll
        for (int h = 0; h < NILA; h++) {
            const int item = FULL * 64 * NH + 64 * h + lane;
            ok[h] = item < NITEM;
            const int it = ok[h] ? item : 0;
            const int sft = it / PRIME;
            kq[h] = it - sft * PRIME + 1;
            so[h] = sft * NDIM;
        }
        acc += eval_items_ps<T, NILA, CFL, K, FL>(kq, so, ok, lat, cf, lm, slab, infi_c, tailq, lane, kk);
    }
    return acc;
}

// The same lattice sum for larger compile-time dimensions (T = 7 .. 16: the Monte-Carlo pattern switch keeps batches of
// up to 16 feasible): the factor (T(T-1)/2 wave-uniform values, 120 at T = 16) no longer fits registers and is read from
// LDS at use (`slab`: packed lower triangle with diagonal, then the limits -- the evaluator's record), the lattice
// coordinates are formed per stage instead of up front; the conditioned values y (2 NH chains x T-1) stay in registers
// because every index is a compile-time constant.
#ifndef ITAL_BIG_HOIST0
// The first variable's interval does not depend on the lattice point (no conditioning yet): its Phi is evaluated once per
// call (qmc_lane_sum_big) instead of once per chain and round -- 1 of T Phi evaluations, 2.6 % (T = 16) .. 6 % (T = 7) of
// the loop's vector instructions.  The compile-time evaluators with the factor in registers get this from the compiler
// (loop-invariant code motion); here the factor is re-read from LDS behind a memory clobber, which pins it inside the loop.
#define ITAL_BIG_HOIST0 1
#endif
#ifndef ITAL_BIG_PIN_FF
#define ITAL_BIG_PIN_FF 1
#endif
#ifndef ITAL_BIG_PIN_LAT
#define ITAL_BIG_PIN_LAT 1
#endif
#ifndef ITAL_BIG_GROUP
// Chains of a lane evaluated together by the scheduler (0: all NCB): Phi and the central Phi^-1 of the NCB chains of a stage
// in groups of this many, with a scheduling barrier between the groups.  All NCB chains interleaved keep NCB sets of
// temporaries live; in groups the lane carries more chains (fuller waves in the Phi^-1 tail branch: 121 instructions per
// pass whatever the number of lanes in it) at the register budget of three waves per SIMD.
#define ITAL_BIG_GROUP(T) 0
#endif
#ifndef ITAL_BIG_ROWGRP
#define ITAL_BIG_ROWGRP 2
#endif
#ifndef ITAL_BIG_YLDS
// Conditioned values of the LAST ITAL_BIG_YLDS(T) stages in LDS instead of registers (`yl`: [YL][NCB][64] doubles of
// wave-private memory): a value written at stage j is read once per later stage, so the latest ones are the cheapest to keep
// there (YL (YL + 1) / 2 reads per chain and point) -- what lets an instantiation fit one more wave per SIMD.  0: none.
#define ITAL_BIG_YLDS(T) ((T) == 14 ? 2 : (T) == 15 ? 4 : (T) == 16 ? 2 : 0)
#endif

template <int T, int NCB, class K, bool FL = false, int YL = 0, int GRP = 0>
__device__ __forceinline__ double eval_chains_big(const int (&kk)[NCB], const int (&so)[NCB], const bool (&anti)[NCB],
                                                  const bool (&ok)[NCB], const double* __restrict__ lat,
                                                  const double* __restrict__ slab, unsigned infi_c, double* tailq, int lane,
                                                  const K& coef, double w0 = 0.0, double* __restrict__ yl = nullptr) {
    constexpr int NDIM = T - 1, NCOV = T * (T + 1) / 2;
    constexpr int NREG = NDIM - YL > 0 ? NDIM - YL : 1;      // stages 0 .. NDIM - YL - 1 keep their value in registers
    double yy[NCB][NREG], ff[NCB];
#pragma unroll
    for (int c = 0; c < NCB; c++) ff[c] = ok[c] ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < T; i++) {
        const bool lower = (infi_c >> i) & 1u;
        double pin[NCB], ph[NCB];
        if (ITAL_BIG_HOIST0 && i == 0) {
#pragma unroll
            for (int c = 0; c < NCB; c++) ph[c] = w0;           // already in its final form (flip_width applied)
        } else {
            __asm__ volatile("" ::: "memory");   // keeps this row's factor loads here: hoisted out of the lattice loop they
                                                 // would occupy T(T+1)/2 register pairs
            const double lmi = slab[NCOV + i];
            double sc[NCB];
#pragma unroll
            for (int c = 0; c < NCB; c++) sc[c] = 0;
#pragma unroll
            for (int j = 0; j < i; j++) {
                // (the factor row is read in groups of ITAL_BIG_ROWGRP values: all i loads of a row issued together would hold
                // 2 i registers until their FMAs retire -- the register peak of the wide instantiations)
                if (ITAL_BIG_ROWGRP > 0 && j > 0 && j % ITAL_BIG_ROWGRP == 0) __asm__ volatile("" ::: "memory");
                const double cij = slab[i * (i + 1) / 2 + j];
#pragma unroll
                for (int c = 0; c < NCB; c++) {
                    const double yj = (YL > 0 && j >= NDIM - YL) ? yl[((j - (NDIM - YL)) * NCB + c) * 64 + lane] : yy[c][j < NREG ? j : 0];
                    sc[c] = fma(cij, yj, sc[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < NCB; c++) {
                if (GRP > 0 && c % GRP == 0) __builtin_amdgcn_sched_barrier(0);
                ph[c] = mvn_phi_lat(lmi - sc[c], coef);
            }
            if (GRP > 0) __builtin_amdgcn_sched_barrier(0);
            if (FL) flip_width<NCB>(ph, lower);
        }
#pragma unroll
        for (int c = 0; c < NCB; c++) {
            const double d = (!FL && lower) ? ph[c] : 0.0;
            const double w = (!FL && lower) ? 1.0 - ph[c] : ph[c];
            ff[c] *= w;
            // (the running product is formed NOW: left alone, the scheduler sinks the T multiplications of a chain to the end
            // of the round and keeps every stage's width alive until then -- 2 registers per chain and stage, 90 of the 244
            // registers of the T = 16 instantiation with three chains)
            if (ITAL_BIG_PIN_FF) __asm__ volatile("" : "+v"(ff[c]));
            if (i < T - 1) {
                // the lattice is read HERE: `lat` is a noalias pointer, so the compiler may (and did) issue the 2 (T - 1)
                // loads of every lattice item at the top of the round, across the clobbers above -- 4 registers per item
                // and stage held to the end of the round (120 of the 244 registers of the T = 16 instantiation).  An
                // offset the compiler cannot see through until this point keeps the loads in their stage.
                int sof = so[c];
                if (ITAL_BIG_PIN_LAT) __asm__ volatile("" : "+v"(sof));
                const double v = kk[c] * lat[sof + i] + lat[8 * NDIM + sof + i];
                const double fr = v - floor(v);
                const double x0 = fabs(2 * fr - 1);
                pin[c] = FL ? (anti[c] ? 1 - x0 : x0) * w : fma(anti[c] ? 1 - x0 : x0, w, d);
            }
        }
        if (i < T - 1) {
            double out[NCB];
            phinv_wave<NCB, K, GRP>(pin, out, tailq, lane, coef);
#pragma unroll
            for (int c = 0; c < NCB; c++) {
                if (YL > 0 && i >= NDIM - YL) yl[((i - (NDIM - YL)) * NCB + c) * 64 + lane] = out[c];   // (own lane only: no fence needed)
                else yy[c][i < NREG ? i : 0] = out[c];
            }
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < NCB; c++) acc += ff[c];
    return acc;
}

// NCB chains per lane and round: whole lattice items (a point and its antithetic partner on the same lane) first; with an
// odd NCB the last chain of lanes 2i and 2i + 1 is the point and the partner of one more item (32 NCB items per round).
template <int T, int NCB, class K = LitK, bool FL = false, int YL = 0, int GRP = 0>
__device__ __forceinline__ double qmc_lane_sum_big(const double* __restrict__ lat, const double* __restrict__ slab,
                                                   unsigned infi_c, double* __restrict__ tailq, int lane, const K& coef = K(),
                                                   double* __restrict__ yl = nullptr) {
    constexpr int NDIM = T - 1, PRIME = P_TAB[(NDIM < 10 ? NDIM : 10) - 1], NCOV = T * (T + 1) / 2;
    constexpr int NITEM = 8 * PRIME, PER_ROUND = 32 * NCB;
    double w0 = 0.0;
    if (ITAL_BIG_HOIST0) {
        // (a variable bounded below outside the all-upper form contributes d = Phi, w = 1 - Phi: the loop forms both from ph)
        double ph0[1] = {mvn_phi_lat(slab[NCOV], coef)};
        if (FL) flip_width<1>(ph0, infi_c & 1u);
        w0 = ph0[0];
    }
    double acc = 0.0;
    for (int base = 0; base < NITEM; base += PER_ROUND) {
        int kk[NCB], so[NCB];
        bool anti[NCB], ok[NCB];
#pragma unroll
        for (int c = 0; c < NCB; c++) {
            const bool paired = c < 2 * (NCB / 2);
            const int item = paired ? base + 64 * (c >> 1) + lane : base + 64 * (NCB / 2) + (lane >> 1);
            anti[c] = paired ? (c & 1) : (lane & 1);
            ok[c] = item < NITEM;
            const int it = ok[c] ? item : 0;
            const int sft = it / PRIME;
            kk[c] = it - sft * PRIME + 1;
            so[c] = sft * NDIM;
        }
        acc += eval_chains_big<T, NCB, K, FL, YL, GRP>(kk, so, anti, ok, lat, slab, infi_c, tailq, lane, coef, w0, yl);
    }
    return acc;
}

}  // namespace ital
