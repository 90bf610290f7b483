// RBF kernel columns + whitened posterior updates, streaming over the rows of X (never N x N).
//
// Roles: R1 RBF kernel (reference ital/gp.py:390-416), R3 posterior mean/variance
// (gp.py:221-232) and R4 batch cross-covariances (gp.py:235-261), in the Cholesky-whitened form
//     G = L L^T,  V = L^-1 K[T,:],  mu = V^T alpha,  s2 = v - colsum(V^2),  Cov(i,b) = K(i,b) - V[:,i].V[:,b].
//
// One kernel does all of it.  For a 16-row tile of X per wave it forms, on the FP64 matrix cores
// (v_mfma_f64_16x16x4_f64),
//     dot[j][i] = <xs_j, x_i>            (the -2 X Xs^T term; K-dimension = d, 16 k-values per step)
//     S[j][i]   = sum_r W[j][r] V[r][i]  (K-dimension = m)
// and then R[j][i] = var * exp((|x_i|^2 + |xs_j|^2 - 2 dot)/(-2 l^2)) - S[j][i] with the row norms fused in
// the epilogue.  Epilogues:
//   MODE_RAW    out[j][i] = R[j][i]                        (kernel columns; greedy cross-covariance column)
//   MODE_WHITEN V_new = L22^-1 R (forward substitution through an LDS transpose), appended as rows m..m+c-1
//               of V; s2 -= colsum(V_new^2); mu += V_new^T alpha_new   (GP update, rank-c Cholesky append)
//   MODE_PREDICT out_mean[i] += ..., used for gp.predict on external points (see predict kernel below)
//
// Data layout: X row-major [n][ldx] fp64, ldx = d padded to a multiple of 16 with zeros (rows 128-B aligned);
// V row-major [m_max][ldv] (one contiguous n-vector per labelled point, coalesced across candidates).
// HBM-bound: algorithmic bytes per row = 8*(d + m) read + 8*c written; the MFMA pipe has ~3x headroom.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "ital_hip.h"
#include "ital_internal.h"
#include "qmc_seed.h"

namespace ital {

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void row_norms_kernel(const double* __restrict__ X, int64_t n, int ldx,
                                                        double* __restrict__ xnorm) {
    // one wave per row, 16-B loads, wave reduction
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const double* row = X + i * ldx;
        double acc = 0;
        for (int k = lane * 2; k < ldx; k += 128) {
            double2 v = *reinterpret_cast<const double2*>(row + k);
            acc += v.x * v.x + v.y * v.y;
        }
        acc = wave_sum(acc);
        if (lane == 0) xnorm[i] = acc;
    }
}

struct KcolsArgs {
    const double* X;      // [n][ldx]
    const double* xnorm;  // [n]
    int64_t n;
    int ldx;              // padded feature dim, multiple of 16
    const double* Xs;     // [c][ldx] selected rows (replicated)
    const double* sn;     // [c] their squared norms
    int c;                // 1..16
    const double* W;      // [c][ldw] coefficients against V rows (L21, or a V column); may be null when m == 0
    int ldw;
    const double* V;      // [m_max][ldv]
    int64_t ldv;
    int m;                // rows of V in use
    double var, s;        // kernel variance; s = -2 l^2
    int mode;
    double* out;          // MODE_RAW: [c][ldo]
    int64_t ldo;
    // MODE_WHITEN
    const double* L22;    // [c][ldw] lower (rows m.. of L, columns m..)
    const double* alpha_new;  // [c]
    double* Vw;           // writable V (same buffer as V)
    double* mu;           // [n]
    double* s2;           // [n]
};

enum { MODE_RAW = 0, MODE_WHITEN = 1 };
#ifndef ITAL_KCOLS_KU
#define ITAL_KCOLS_KU 4
#endif
constexpr int KU = ITAL_KCOLS_KU;   // feature steps per trip of the dot-product loop

__device__ __forceinline__ void kcols_body(const KcolsArgs& a) {
    __shared__ double tile[4][16][17];  // per wave: R[j][i] transpose buffer for the whitening epilogue
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col = lane & 15;   // MFMA: A row / B column / D column
    const int kg = lane >> 4;    // MFMA: k index / D row group
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
    if (i0 >= a.n) return;
    const int64_t irow = i0 + col;            // data row this lane feeds as B column
    const bool row_ok = irow < a.n;
    const bool sel_ok = col < a.c;            // selected point this lane feeds as A row
    const double* xrow = a.X + (row_ok ? irow : 0) * a.ldx;
    const double* srow = a.Xs + (sel_ok ? col : 0) * (int64_t)a.ldx;

    d4 acc_dot = {0, 0, 0, 0};
    // ---- dot products over the feature dimension: 16 k-values per step, 32 B per lane per operand
    // KU steps of 16 features per trip, all their loads issued before the first MFMA: at the sizes where the kernel is
    // latency bound (a few hundred waves) the operand fetches of a trip overlap instead of queueing behind each other.
    for (int k0 = 0; k0 < a.ldx; k0 += 16 * KU) {
        double2 b01[KU], b23[KU], a01[KU], a23[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int kk = k0 + 16 * u + 4 * kg;
            const bool in = k0 + 16 * u < a.ldx;           // ldx is a multiple of 16
            b01[u] = b23[u] = a01[u] = a23[u] = double2{0, 0};
            if (in && row_ok) {
                b01[u] = *reinterpret_cast<const double2*>(xrow + kk);
                b23[u] = *reinterpret_cast<const double2*>(xrow + kk + 2);
            }
            if (in && sel_ok) {
                a01[u] = *reinterpret_cast<const double2*>(srow + kk);
                a23[u] = *reinterpret_cast<const double2*>(srow + kk + 2);
            }
        }
#pragma unroll
        for (int u = 0; u < KU; u++) {
            acc_dot = __builtin_amdgcn_mfma_f64_16x16x4f64(a01[u].x, b01[u].x, acc_dot, 0, 0, 0);
            acc_dot = __builtin_amdgcn_mfma_f64_16x16x4f64(a01[u].y, b01[u].y, acc_dot, 0, 0, 0);
            acc_dot = __builtin_amdgcn_mfma_f64_16x16x4f64(a23[u].x, b23[u].x, acc_dot, 0, 0, 0);
            acc_dot = __builtin_amdgcn_mfma_f64_16x16x4f64(a23[u].y, b23[u].y, acc_dot, 0, 0, 0);
        }
    }
    // ---- S = W V over the labelled dimension: A[j][r] = W[j][r], B[r][i] = V[r][i]
    d4 acc_s = {0, 0, 0, 0};
    for (int r0 = 0; r0 < a.m; r0 += 16) {
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = r0 + 4 * u + kg;
            av[u] = bv[u] = 0;
            if (r < a.m) {
                if (sel_ok) av[u] = a.W[(int64_t)col * a.ldw + r];
                if (row_ok) bv[u] = a.V[(int64_t)r * a.ldv + irow];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc_s = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc_s, 0, 0, 0);
    }
    // ---- epilogue.  D layout (f64 16x16x4): element reg -> row j = kg + 4*reg, column i = col.
    const double xn = row_ok ? a.xnorm[irow] : 0.0;
    double R[4];
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int j = kg + 4 * reg;
        double v = 0;
        if (j < a.c) {
            const double snj = a.sn[j];
            v = a.var * exp((snj + xn - 2 * acc_dot[reg]) / a.s) - acc_s[reg];
        }
        R[reg] = v;
    }
    if (a.mode == MODE_RAW) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int j = kg + 4 * reg;
            if (j < a.c && row_ok) a.out[(int64_t)j * a.ldo + irow] = R[reg];
        }
        return;
    }
    // MODE_WHITEN: transpose through LDS so that one lane owns all c values of a data row
#pragma unroll
    for (int reg = 0; reg < 4; reg++) tile[wave][kg + 4 * reg][col] = R[reg];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 16 && row_ok) {
        double vn[16];
        double dvar = 0, dmu = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j < a.c) {
                double acc = tile[wave][j][lane];
#pragma unroll
                for (int q = 0; q < j; q++) acc -= a.L22[j * a.ldw + q] * vn[q];
                vn[j] = acc / a.L22[j * a.ldw + j];
                dvar += vn[j] * vn[j];
                dmu += vn[j] * a.alpha_new[j];
                a.Vw[(int64_t)(a.m + j) * a.ldv + irow] = vn[j];
            } else {
                vn[j] = 0;
            }
        }
        a.s2[irow] -= dvar;
        a.mu[irow] += dmu;
    }
}

__global__ __launch_bounds__(256) void kcols_kernel(KcolsArgs a) { kcols_body(a); }

// The covariance column of the member just selected AND the generator states of the next greedy step's candidates in one
// launch: both depend on nothing but that selection, both are short (10 us each at 9298 rows) -- the first `nk` workgroups
// stream the column, the rest compute the seeds (qmc_seed.h) on other CUs at the same time.
__global__ __launch_bounds__(256) void kcols_seed_kernel(KcolsArgs a, SeedArgs s, unsigned nk) {
    if (blockIdx.x >= nk) {
        qmc_seed_body(s, (int64_t)(blockIdx.x - nk) * 256 + threadIdx.x);
        return;
    }
    kcols_body(a);
}

// gp.predict on external points (reference ital/gp.py:264-292): the test points are whitened against the labelled set
// exactly as the data rows are -- 16 labelled points per sweep of kcols_kernel in MODE_WHITEN (-2 Xt XT^T on FP64 MFMA,
// forward substitution against the Cholesky factor, mean += V^T alpha, var -= colsum(V^2)) -- which also leaves the
// whitened columns Vt for the full predictive covariance K(Xt, Xt) - Vt^T Vt (ital_cov_block).
__global__ __launch_bounds__(256) void predict_init_kernel(int64_t nt, double var, double* __restrict__ mean,
                                                           double* __restrict__ pvar) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt) { mean[i] = 0.0; pvar[i] = var; }
}

__global__ __launch_bounds__(256) void clamp0_kernel(int64_t nt, double* __restrict__ pvar) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt) pvar[i] = fmax(0.0, pvar[i]);     // np.maximum(0, ...) of gp.py:290
}

// Stages the samples an update labels: feature rows picked out of a row matrix (the replicated batch state of the last
// fetch) into the labelled-set rows XT[m .. m+c), their squared norms and the labels -- one launch, the picks and labels
// travel as kernel arguments (no host-to-device copies, no gather / copy / norm launches of their own).
__global__ __launch_bounds__(64) void stage_labelled_kernel(const double* __restrict__ rows, int ldx, ital_label_batch lb,
                                                            double* __restrict__ XT_dst, double* __restrict__ XTn_dst,
                                                            double* __restrict__ y_dst) {
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= lb.c) return;
    const double* src = rows + (int64_t)lb.slot[j] * ldx;
    double* dst = XT_dst + (int64_t)j * ldx;
    double acc = 0;
    for (int k = lane; k < ldx; k += 64) {
        const double v = src[k];
        dst[k] = v;
        acc += v * v;
    }
    acc = wave_sum(acc);
    if (lane == 0) { XTn_dst[j] = acc; y_dst[j] = lb.y[j]; }
}

}  // namespace ital

using namespace ital;

extern "C" int ital_stage_labelled(const double* rows, int ldx, ital_label_batch lb, double* XT_dst, double* XTn_dst,
                                   double* y_dst, hipStream_t stream) {
    if (lb.c < 1 || lb.c > 16) return ital_fail(-22, "ital_stage_labelled: 1..16 samples per call");
    if (ldx % 16 != 0) return ital_fail(-22, "ital_stage_labelled: ldx must be a multiple of 16");
    ITAL_LAUNCH(stage_labelled_kernel, dim3(lb.c), dim3(64), 0, stream, rows, ldx, lb, XT_dst, XTn_dst, y_dst);
    return ital_check_launch("ital_stage_labelled");
}

extern "C" int ital_row_norms(const double* X, int64_t n, int ldx, double* xnorm, hipStream_t stream) {
    if (n <= 0) return 0;
    if (ldx % 16 != 0) return ital_fail(-22, "ital_row_norms: ldx must be a multiple of 16");
    int64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    ITAL_LAUNCH(row_norms_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, X, n, ldx, xnorm);
    return ital_check_launch("ital_row_norms");
}

static int launch_kcols(KcolsArgs& a, hipStream_t stream, const char* who) {
    if (a.n <= 0) return 0;
    if (a.ldx % 16 != 0) return ital_fail(-22, "kcols: ldx must be a multiple of 16");
    if (a.c < 1 || a.c > 16) return ital_fail(-22, "kcols: c must be in 1..16");
    if (a.m < 0 || (a.m > 0 && (!a.W || !a.V))) return ital_fail(-22, "kcols: W/V missing");
    int64_t blocks = (a.n + 63) / 64;
    ITAL_LAUNCH(kcols_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return ital_check_launch(who);
}

extern "C" int ital_rbf_cols(const double* X, const double* xnorm, int64_t n, int ldx, const double* Xs,
                             const double* sn, int c, double var, double length_scale, double* out, int64_t ldo,
                             hipStream_t stream) {
    KcolsArgs a = {};
    a.X = X; a.xnorm = xnorm; a.n = n; a.ldx = ldx; a.Xs = Xs; a.sn = sn; a.c = c;
    a.m = 0; a.var = var; a.s = -2.0 * length_scale * length_scale; a.mode = MODE_RAW; a.out = out; a.ldo = ldo;
    return launch_kcols(a, stream, "ital_rbf_cols");
}

extern "C" int ital_cross_cov_cols(const double* X, const double* xnorm, int64_t n, int ldx, const double* Xs,
                                   const double* sn, int c, const double* W, int ldw, const double* V, int64_t ldv,
                                   int m, double var, double length_scale, double* out, int64_t ldo,
                                   hipStream_t stream) {
    KcolsArgs a = {};
    a.X = X; a.xnorm = xnorm; a.n = n; a.ldx = ldx; a.Xs = Xs; a.sn = sn; a.c = c;
    a.W = W; a.ldw = ldw; a.V = V; a.ldv = ldv; a.m = m;
    a.var = var; a.s = -2.0 * length_scale * length_scale; a.mode = MODE_RAW; a.out = out; a.ldo = ldo;
    return launch_kcols(a, stream, "ital_cross_cov_cols");
}

// ital_cross_cov_cols with the seed computation of the next greedy step riding along (round driver, round.hip).
int ital_cross_cov_cols_seed(const double* X, const double* xnorm, int64_t n, int ldx, const double* Xs, const double* sn, int c,
                             const double* W, int ldw, const double* V, int64_t ldv, int m, double var, double length_scale,
                             double* out, int64_t ldo, const ital::SeedArgs& seed, hipStream_t stream) {
    KcolsArgs a = {};
    a.X = X; a.xnorm = xnorm; a.n = n; a.ldx = ldx; a.Xs = Xs; a.sn = sn; a.c = c;
    a.W = W; a.ldw = ldw; a.V = V; a.ldv = ldv; a.m = m;
    a.var = var; a.s = -2.0 * length_scale * length_scale; a.mode = MODE_RAW; a.out = out; a.ldo = ldo;
    if (a.n <= 0 || seed.slab_n <= 0) return ital_fail(-22, "ital_cross_cov_cols_seed: nothing to do");
    if (a.ldx % 16 != 0) return ital_fail(-22, "kcols: ldx must be a multiple of 16");
    if (a.c < 1 || a.c > 16) return ital_fail(-22, "kcols: c must be in 1..16");
    if (a.m < 0 || (a.m > 0 && (!a.W || !a.V))) return ital_fail(-22, "kcols: W/V missing");
    const int64_t nk = (a.n + 63) / 64, ns = (seed.slab_n + 255) / 256;
    ITAL_LAUNCH(kcols_seed_kernel, dim3((unsigned)(nk + ns)), dim3(256), 0, stream, a, seed, (unsigned)nk);
    return ital_check_launch("ital_cross_cov_cols_seed");
}

extern "C" int ital_whiten_append(const double* X, const double* xnorm, int64_t n, int ldx, const double* Xnew,
                                  const double* snew, int c, const double* L21, int ldw, const double* L22,
                                  const double* alpha_new, double* V, int64_t ldv, int m, double var,
                                  double length_scale, double* mu, double* s2, hipStream_t stream) {
    KcolsArgs a = {};
    a.X = X; a.xnorm = xnorm; a.n = n; a.ldx = ldx; a.Xs = Xnew; a.sn = snew; a.c = c;
    a.W = L21; a.ldw = ldw; a.V = V; a.Vw = V; a.ldv = ldv; a.m = m;
    a.var = var; a.s = -2.0 * length_scale * length_scale; a.mode = MODE_WHITEN;
    a.L22 = L22; a.alpha_new = alpha_new; a.mu = mu; a.s2 = s2;
    return launch_kcols(a, stream, "ital_whiten_append");
}

extern "C" int ital_predict(const double* Xt, int64_t nt, int ldx, const double* XT, const double* XTn, int m,
                            const double* L, int ldl, const double* alpha, double var, double length_scale,
                            double* mean, double* pvar, double* xtn, double* Vt, int64_t ldvt, int clamp,
                            hipStream_t stream) {
    if (nt <= 0) return 0;
    if (m <= 0) return ital_fail(-22, "ital_predict: GP is not fitted");
    if (!mean || !pvar || !xtn || !Vt) return ital_fail(-22, "ital_predict: mean / pvar / xtn / Vt must all be given");
    if (ldvt < nt) return ital_fail(-22, "ital_predict: ldvt smaller than the number of test points");
    int rc = ital_row_norms(Xt, nt, ldx, xtn, stream);
    if (rc) return rc;
    const unsigned blocks = (unsigned)((nt + 255) / 256);
    ITAL_LAUNCH(predict_init_kernel, dim3(blocks), dim3(256), 0, stream, nt, var, mean, pvar);
    for (int c0 = 0; c0 < m; c0 += 16) {
        const int c = m - c0 < 16 ? m - c0 : 16;
        rc = ital_whiten_append(Xt, xtn, nt, ldx, XT + (int64_t)c0 * ldx, XTn + c0, c, L + (int64_t)c0 * ldl, ldl,
                                L + (int64_t)c0 * ldl + c0, alpha + c0, Vt, ldvt, c0, var, length_scale, mean, pvar, stream);
        if (rc) return rc;
    }
    if (clamp) ITAL_LAUNCH(clamp0_kernel, dim3(blocks), dim3(256), 0, stream, nt, pvar);
    return ital_check_launch("ital_predict");
}
