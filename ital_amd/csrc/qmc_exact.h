// MVKBRV's own summation order for the orthant calls where it decides something (label_estimation 'optimistic' /
// 'pessimistic', reference ital/ital.py:210-216: `if (mi == 0) or (cur_mi < mi)` resets the running value on EXACT
// equality, and cur_mi = log(pu + eps) - log(pr + eps) is exactly 0 when a pattern's prior probability equals its updated
// one bit for bit -- e.g. both 1).  Genz's DKBVRC / DKSMRC (scipy/stats/mvndst.f, restated in oracle/mvndst_oracle.c) form
// the estimate as a serial running mean over the 2 P values of a shifted lattice (a point, its antithetic partner, the next
// point ...), then a running mean over the 8 shifts; the lattice-sum kernels add the 16 P values up in parallel and divide
// once.  Both are the same number to ~1e-16 -- but where the serial recurrence leaves 1 - 2e-16 and the flat sum rounds to
// 1, the reference's equality test goes the other way.  Calls whose flat sum lands within 1e-9 of 0 or 1 are therefore
// flagged by the lattice-sum kernels (only with those label modes) and recomputed here: every point's value with MVNDFN's
// own (un-negated) arithmetic into LDS, then the two recurrences exactly as the reference runs them.
#pragma once
#include <hip/hip_runtime.h>

#include "qmc_common.h"

namespace ital {

constexpr double EXACT_BAND = 1e-9;      // a flat sum this close to 0 or 1 is recomputed in the reference's order

// The value of every lattice point of a prepared call of dimension n <= NMAX (natural form: limit types in `infi`, rows
// that close a group of MVNDFN in `closes`): vals[shift][2 (k - 1) + c], c = 0 the point k, c = 1 its antithetic partner --
// the order DKSMRC visits them in.  slab: packed lower factor with diagonal, then the limits; lat: [8][n-1] permuted
// generators, then [8][n-1] shifts; tailq: 128 doubles of wave-private LDS.
template <int NMAX>
__device__ void qmc_point_values(int n, const double* __restrict__ slab, unsigned infi, unsigned closes,
                                 const double* __restrict__ lat, int lane, double* __restrict__ tailq, double* __restrict__ vals) {
    const int ndim = n - 1;
    const int prime = P_TAB[(ndim < 10 ? ndim : 10) - 1];
    const double* cf = slab;
    const double* lm = slab + n * (n + 1) / 2;
    const int items = 8 * prime;
    for (int base = 0; base < items; base += 64) {
        double yy[2][NMAX - 1], ff[2], ai[2], bi[2];
        bool dead[2];
        const int item = base + lane;
        const bool ok = item < items;
        const int it = ok ? item : 0;
        const int sft = it / prime;
        const int kk = it - sft * prime + 1;
        const int so = sft * ndim;
        ff[0] = ff[1] = 1.0;
        dead[0] = dead[1] = !ok;
        ai[0] = ai[1] = bi[0] = bi[1] = 0;
        bool infa = false, infb = false;   // wave-uniform: the open group has a lower / an upper limit (MVNDFN)
        int ik = 0;                        // groups closed so far = lattice coordinate of the open group
#pragma unroll
        for (int i = 0; i < NMAX; i++) {
            if (i < n) {   // uniform; no `break`: the body holds convergent wave operations
                const bool lower = (infi >> i) & 1u;
                const bool close = (closes >> i) & 1u;
                const bool last = i == n - 1;
                const double lmi = lm[i];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    double sc = 0;
#pragma unroll
                    for (int j = 0; j < i; j++) sc = fma(cf[pidx(i, j)], yy[c][j], sc);
                    const double z = lmi - sc;
                    if (lower) ai[c] = infa ? fmax(ai[c], z) : z;
                    else bi[c] = infb ? fmin(bi[c], z) : z;
                }
                if (lower) infa = true; else infb = true;
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
#pragma unroll
                        for (int c = 0; c < 2; c++)
                            if (i < NMAX - 1) yy[c][i < NMAX - 1 ? i : 0] = outv[c];
                    }
                    infa = false; infb = false;
                    ik++;
                } else {
#pragma unroll
                    for (int c = 0; c < 2; c++)
                        if (i < NMAX - 1) yy[c][i < NMAX - 1 ? i : 0] = 0.0;
                }
            }
        }
        if (ok) {
            vals[sft * 2 * prime + 2 * (kk - 1)] = dead[0] ? 0.0 : ff[0];
            vals[sft * 2 * prime + 2 * (kk - 1) + 1] = dead[1] ? 0.0 : ff[1];
        }
    }
}

// DKSMRC's running mean over the 2 P values of each shift (lanes 0 .. 7, one shift each), then DKBVRC's running mean over
// the 8 shifts; MVNDST's estimate after its single pass (FINEST = 0 + (FINVAL - 0) / (1 + 0)).  The same value in every lane.
__device__ __forceinline__ double mvkbrv_serial(int prime, const double* __restrict__ vals, int lane) {
    double sumkro = 0.0;
    if (lane < 8) {
        const double* v = vals + lane * 2 * prime;
        for (int k = 1; k <= prime; k++) {
            sumkro = sumkro + (v[2 * k - 2] - sumkro) / (double)(2 * k - 1);
            sumkro = sumkro + (v[2 * k - 1] - sumkro) / (double)(2 * k);
        }
    }
    double finval = 0.0;
    for (int i = 1; i <= 8; i++) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(sumkro), i - 1);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(sumkro), i - 1);
        const double value = __hiloint2double(hi, lo);
        const double difint = (value - finval) / (double)i;
        finval = finval + difint;
    }
    return finval;
}

__device__ __forceinline__ void exact_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MVKBRV's estimate of a prepared call of ANY dimension n (runtime; <= ystride + 1) in the reference's own summation order,
// with a constant amount of LDS -- what the general scorer's single kernel and its pipeline above 8 variables use (round 6:
// 16 P values per call are 174 KB at P = 1361).  The lattice points are visited in blocks of 8 per shift: lane = 8 sft + q
// evaluates point k0 + q + 1 of shift sft and its antithetic partner with MVNDFN's own arithmetic (the operations of
// qmc_eval_lds / qmc_point_values in their order; the chains' conditioned values in `yl`: [2][ystride][64] doubles), the 128
// values go to `vals`, then lanes 0 .. 7 advance DKSMRC's running mean of their shift by the block's 16 values -- the same
// dependent steps as mvkbrv_serial, interleaved with the evaluation instead of behind it.  DKBVRC's mean over the shifts at
// the end.  The same value in every lane.  tailq: 128 doubles, vals: 128 doubles of wave-private LDS.
static __device__ __attribute__((noinline)) double qmc_exact_lds(int n, const double* __restrict__ slab, unsigned infi, unsigned closes,
                                                                const double* __restrict__ lat, int lane, double* __restrict__ tailq,
                                                                double* __restrict__ yl, int ystride, double* __restrict__ vals) {
    const int ndim = n - 1;
    const int prime = P_TAB[(ndim < 10 ? ndim : 10) - 1];
    const double* cf = slab;
    const double* lm = slab + n * (n + 1) / 2;
    const int sft = lane >> 3, kq = lane & 7;
    const int so = sft * ndim;
    double* y0 = yl + lane;
    double* y1 = yl + ystride * 64 + lane;
    double sumkro = 0.0;
    for (int k0 = 0; k0 < prime; k0 += 8) {
        const bool ok = k0 + kq < prime;
        const int kk = ok ? k0 + kq + 1 : 1;
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
        vals[sft * 16 + 2 * kq] = dead[0] ? 0.0 : ff[0];
        vals[sft * 16 + 2 * kq + 1] = dead[1] ? 0.0 : ff[1];
        exact_wave_sync();
        if (lane < 8) {
            const double* v = vals + lane * 16;
            const int cnt = prime - k0 < 8 ? prime - k0 : 8;
            for (int q = 0; q < cnt; q++) {
                const int k = k0 + q + 1;
                sumkro = sumkro + (v[2 * q] - sumkro) / (double)(2 * k - 1);
                sumkro = sumkro + (v[2 * q + 1] - sumkro) / (double)(2 * k);
            }
        }
        exact_wave_sync();
    }
    double finval = 0.0;
    for (int i = 1; i <= 8; i++) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(sumkro), i - 1);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(sumkro), i - 1);
        const double value = __hiloint2double(hi, lo);
        const double difint = (value - finval) / (double)i;
        finval = finval + difint;
    }
    return finval;
}

}  // namespace ital
