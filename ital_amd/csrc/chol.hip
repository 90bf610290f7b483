// Rank-c Cholesky append of the labelled-set Gram matrix (role R2, reference ital/gp.py:8-37 via :157,:194).
//
// The reference re-inverts K[T,T] + noise*I from scratch at every update; here the lower Cholesky factor L
// (row-major [m_max][ldl]) grows by c rows:
//     L21 = K[new,T] L11^-T,   L22 = chol(K[new,new] + noise*I - L21 L21^T),   alpha_new = L22^-1 (y_new - L21 alpha)
// One workgroup of 16 waves, one wave per new row (c <= 16) after the kernel values: tiny (m <= a few hundred), latency-bound, stays in L2 -- the
// structure below keeps the number of dependent memory round trips per append at O(m / 64), not O(m).
// The new feature rows must already sit in XT rows m..m+c-1 (with their squared norms in XTn).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "ital_hip.h"
#include "ital_internal.h"

namespace ital {

__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Optional staging in front (ital_gp_append): the new samples' feature rows are picked out of a row matrix (the batch state
// of the last fetch), their squared norms and the labels written -- what ital_stage_labelled does in a launch of its own.
struct StageArgs {
    const double* rows;      // nullptr: XT / XTn / ynew are ready
    ital_label_batch lb;
    double* XT_w;            // writable aliases of XT + m * ldx, XTn + m, ynew
    double* XTn_w;
    double* y_w;
};

__global__ __launch_bounds__(1024) void chol_append_kernel(const double* XT, const double* XTn,
                                                           int ldx, double* __restrict__ L, int ldl,
                                                           double* __restrict__ alpha, const double* ynew,
                                                           int m, int c, double var, double s, double noise,
                                                           int* __restrict__ status, StageArgs sa) {
    __shared__ double S[16][17];
    __shared__ double tvec[16];
    __shared__ double Lb[64][65];    // one 64 x 64 diagonal block of the existing factor at a time
    const int lane = threadIdx.x & 63;
    const int j = threadIdx.x >> 6;  // new row handled by this wave
    const int g = m + j;             // its row in L
    double* Lg = L + (int64_t)g * ldl;
    if (sa.rows) {
        if (j < c) {
            const double* src = sa.rows + (int64_t)sa.lb.slot[j] * ldx;
            double* dst = sa.XT_w + (int64_t)j * ldx;
            double acc = 0;
            for (int k = lane; k < ldx; k += 64) {
                const double v = src[k];
                dst[k] = v;
                acc += v * v;
            }
            acc = wave_sum(acc);
            if (lane == 0) { sa.XTn_w[j] = acc; sa.y_w[j] = sa.lb.y[j]; }
        }
        __threadfence_block();
        __syncthreads();
    }
#ifdef ITAL_CHOL_TIMING
    long long tstamp[6];
    tstamp[0] = __builtin_readcyclecounter();
#define CHOL_STAMP(i) tstamp[i] = __builtin_readcyclecounter()
#else
#define CHOL_STAMP(i)
#endif
    {
        // kernel values of the new rows against rows 0..g (written into L as workspace).  The step is bound by memory
        // latency and by cache lines per load, so all 1024 threads of the workgroup take part: sixteen adjacent lanes share
        // one (row, new row) pair and read 256 contiguous bytes of either row per load, up to eight such loads per row in
        // flight (256 features per step), and a sixteen-lane sum replaces a 64-lane one.
        const int sub = threadIdx.x & 15;
        const int npair = (m + c) * c;
        for (int p0 = 0; p0 < npair; p0 += (int)blockDim.x >> 4) {
            const int pr = p0 + ((int)threadIdx.x >> 4);
            const bool live = pr < npair;
            const int jn = live ? pr % c : 0, r = live ? pr / c : 0;
            const int gn = m + jn;
            const bool need = live && r <= gn;
            double d0 = 0, d1 = 0;
            if (need) {
                const double* xa = XT + (int64_t)gn * ldx + 2 * sub;
                const double* xb = XT + (int64_t)r * ldx + 2 * sub;
                for (int k0 = 0; k0 < ldx; k0 += 256) {
                    double2 a[8], b[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int k = k0 + 32 * u;
                        const bool in = k + 2 * sub < ldx;           // ldx is a multiple of 16: pairs never straddle the end
                        a[u] = in ? *reinterpret_cast<const double2*>(xa + k) : double2{0, 0};
                        b[u] = in ? *reinterpret_cast<const double2*>(xb + k) : double2{0, 0};
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        d0 = fma(a[u].x, b[u].x, d0);
                        d1 = fma(a[u].y, b[u].y, d1);
                    }
                }
            }
            double dot = d0 + d1;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) dot += __shfl_xor(dot, off, 64);
            if (need && sub == 0) {
                double kv = var * exp((XTn[gn] + XTn[r] - 2 * dot) / s);
                if (r == gn) kv += noise;
                L[(int64_t)gn * ldl + r] = kv;
            }
        }
    }
    CHOL_STAMP(1);
    // L21 rows: forward substitution against the existing factor, 64 columns at a time.  The diagonal block is staged
    // in LDS once for all new rows; lane i of wave j carries unknown r0 + i of row j, and the recurrence runs over the
    // columns (the solved value of column q is broadcast, every later lane subtracts its multiple) -- m short steps
    // instead of m dot products with a reduction each.
    for (int r0 = 0; r0 < m; r0 += 64) {
        const int nb = m - r0 < 64 ? m - r0 : 64;
        __syncthreads();             // the previous block is no longer read; this wave's kernel values are written
        for (int idx = threadIdx.x; idx < nb * nb; idx += blockDim.x) {
            const int i = idx / nb, q = idx - i * nb;
            Lb[i][q] = L[(int64_t)(r0 + i) * ldl + r0 + q];
        }
        __syncthreads();
        if (j < c) {
            const bool in = lane < nb;
            double b = 0.0;
            if (in) {
                b = Lg[r0 + lane];
                const double* Li = L + (int64_t)(r0 + lane) * ldl;
                for (int q = 0; q < r0; q++) b = fma(-Li[q], Lg[q], b);     // columns solved in earlier blocks
            }
            const double dinv = in ? 1.0 / Lb[lane][lane] : 1.0;       // one division for the whole block
            double lq = Lb[lane][0];
            for (int q = 0; q < nb; q++) {
                const double lnext = Lb[lane][q + 1 < nb ? q + 1 : q];   // next column's multiplier, off the dependent chain
                const double xq = readlane_f64(b * dinv, q);              // lane q's own product, broadcast
                if (lane == q) b = xq;
                else if (lane > q && in) b = fma(-lq, xq, b);
                lq = lnext;
            }
            if (in) Lg[r0 + lane] = b;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    CHOL_STAMP(2);
    __threadfence_block();
    __syncthreads();
    if (j < c) {
        // Schur complement row j: S[j][q] = K[g][m+q] - L21[j] . L21[q], q <= j ; t_j = y_j - L21[j] . alpha
        for (int q = 0; q <= j; q++) {
            const double* Lq = L + (int64_t)(m + q) * ldl;
            double acc = 0;
            for (int r = lane; r < m; r += 64) acc += Lg[r] * Lq[r];
            acc = wave_sum(acc);
            if (lane == 0) S[j][q] = Lg[m + q] - acc;
        }
        double acc = 0;
        for (int r = lane; r < m; r += 64) acc += Lg[r] * alpha[r];
        acc = wave_sum(acc);
        if (lane == 0) tvec[j] = ynew[j] - acc;
    }
    CHOL_STAMP(3);
    __syncthreads();
    if (j == 0) {
        // dense c x c Cholesky + forward substitution in wave 0, lane a = row a (c <= 16); right-looking, so that every
        // entry still receives its corrections in the order q = 0, 1, ... of the row-by-row formulation
        const bool mine = lane < c;
        bool bad = false;
        for (int b = 0; b < c; b++) {
            const double dbb = S[b][b];
            if (!(dbb > 0)) bad = true;
            const double sq = sqrt(dbb);
            double lab = 0;
            if (mine && lane > b) {
                lab = S[lane][b] / sq;
                S[lane][b] = lab;
            }
            if (lane == b) S[b][b] = sq;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (mine && lane > b)
                for (int b2 = b + 1; b2 <= lane; b2++) S[lane][b2] -= lab * S[b2][b];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        double tv = mine ? tvec[lane] : 0.0;
        const double dd = mine ? S[lane][lane] : 1.0;
        for (int q = 0; q < c; q++) {
            const double xq = readlane_f64(tv, q) / readlane_f64(dd, q);
            if (lane == q) tv = xq;
            else if (mine && lane > q) tv -= S[lane][q] * xq;
        }
        if (mine) {
            for (int b = 0; b < c; b++) L[(int64_t)(m + lane) * ldl + m + b] = b <= lane ? S[lane][b] : 0.0;
            alpha[m + lane] = tv;
        }
        if (bad && lane == 0) atomicOr(status, 1);  // Gram matrix not positive definite
    }
#ifdef ITAL_CHOL_TIMING
    CHOL_STAMP(4);
    if (threadIdx.x == 0)
        for (int i = 0; i < 4; i++) alpha[ldl - 8 + i] = (double)(tstamp[i + 1] - tstamp[i]);
#endif
}

}  // namespace ital

extern "C" int ital_chol_append(const double* XT, const double* XTn, int ldx, double* L, int ldl, double* alpha,
                                const double* ynew, int m, int c, double var, double length_scale, double noise,
                                int* status, hipStream_t stream) {
    if (c < 1 || c > 16) return ital_fail(-22, "ital_chol_append: c must be in 1..16");
    if (m < 0 || m + c > ldl) return ital_fail(-22, "ital_chol_append: factor capacity exceeded");
    ITAL_LAUNCH(ital::chol_append_kernel, dim3(1), dim3(1024), 0, stream, XT, XTn, ldx, L, ldl, alpha, ynew, m,
                       c, var, -2.0 * length_scale * length_scale, noise, status, ital::StageArgs{});
    return ital_check_launch("ital_chol_append");
}

// update() of the retrieval loop as ONE call (reference ital/gp.py:164-200 + predict_stored, gp.py:203-232): the samples'
// rows staged out of `rows`, the rank-c Cholesky append (both in one single-workgroup launch) and the whitening sweep.
extern "C" int ital_gp_append(const ital_append_desc* a, hipStream_t stream) {
    if (!a) return ital_fail(-22, "ital_gp_append: null descriptor");
    const int c = a->lb.c, m = a->m;
    if (c < 1 || c > 16) return ital_fail(-22, "ital_gp_append: 1..16 samples per call");
    if (m < 0 || m + c > a->ldl) return ital_fail(-22, "ital_gp_append: factor capacity exceeded");
    if (a->ldx % 16 != 0) return ital_fail(-22, "ital_gp_append: ldx must be a multiple of 16");
    ital::StageArgs sa = {a->rows, a->lb, a->XT + (int64_t)m * a->ldx, a->XTn + m, a->ybuf};
    ITAL_LAUNCH(ital::chol_append_kernel, dim3(1), dim3(1024), 0, stream, a->XT, a->XTn, a->ldx, a->L, a->ldl, a->alpha,
                a->ybuf, m, c, a->var, -2.0 * a->length_scale * a->length_scale, a->noise, a->status, sa);
    const int rc = ital_check_launch("ital_gp_append(chol)");
    if (rc) return rc;
    const double* L21 = a->L + (int64_t)m * a->ldl;
    return ital_whiten_append(a->X, a->xnorm, a->n, a->ldx, a->XT + (int64_t)m * a->ldx, a->XTn + m, c, L21, a->ldl, L21 + m,
                              a->alpha + m, a->V, a->ldv, m, a->var, a->length_scale, a->mu, a->s2, stream);
}
