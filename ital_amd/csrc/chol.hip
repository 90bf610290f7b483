// Rank-c Cholesky append of the labelled-set Gram matrix (role R2, reference ital/gp.py:8-37 via :157,:194).
//
// The reference re-inverts K[T,T] + noise*I from scratch at every update; here the lower Cholesky factor L
// (row-major [m_max][ldl]) grows by c rows:
//     L21 = K[new,T] L11^-T,   L22 = chol(K[new,new] + noise*I - L21 L21^T),   alpha_new = L22^-1 (y_new - L21 alpha)
// One workgroup, one wave per new row (c <= 16): tiny (m <= a few hundred), latency-bound, stays in L2.
// The new feature rows must already sit in XT rows m..m+c-1 (with their squared norms in XTn).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "ital_internal.h"

namespace ital {

__global__ __launch_bounds__(1024) void chol_append_kernel(const double* __restrict__ XT, const double* __restrict__ XTn,
                                                           int ldx, double* __restrict__ L, int ldl,
                                                           double* __restrict__ alpha, const double* __restrict__ ynew,
                                                           int m, int c, double var, double s, double noise,
                                                           int* __restrict__ status) {
    __shared__ double S[16][17];
    __shared__ double tvec[16];
    const int lane = threadIdx.x & 63;
    const int j = threadIdx.x >> 6;  // new row handled by this wave
    const int g = m + j;             // its row in L
    if (j < c) {
        // kernel values of row g against rows 0..g (written into L as workspace)
        const double* xg = XT + (int64_t)g * ldx;
        for (int r = 0; r <= g; r++) {
            const double* xr = XT + (int64_t)r * ldx;
            double dot = 0;
            for (int k = lane; k < ldx; k += 64) dot += xg[k] * xr[k];
            dot = wave_sum(dot);
            if (lane == 0) {
                double kv = var * exp((XTn[g] + XTn[r] - 2 * dot) / s);
                if (r == g) kv += noise;
                L[(int64_t)g * ldl + r] = kv;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // L21 row: forward substitution against the existing factor
        double* Lg = L + (int64_t)g * ldl;
        for (int r = 0; r < m; r++) {
            const double* Lr = L + (int64_t)r * ldl;
            double acc = 0;
            for (int q = lane; q < r; q += 64) acc += Lg[q] * Lr[q];
            acc = wave_sum(acc);
            if (lane == 0) Lg[r] = (Lg[r] - acc) / Lr[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    __threadfence_block();
    __syncthreads();
    if (j < c) {
        // Schur complement row j: S[j][q] = K[g][m+q] - L21[j] . L21[q], q <= j ; t_j = y_j - L21[j] . alpha
        const double* Lg = L + (int64_t)g * ldl;
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
    __syncthreads();
    if (threadIdx.x == 0) {
        // dense c x c Cholesky + forward substitution, serial (c <= 16)
        bool bad = false;
        for (int a = 0; a < c; a++) {
            for (int b = 0; b <= a; b++) {
                double v = S[a][b];
                for (int q = 0; q < b; q++) v -= S[a][q] * S[b][q];
                if (a == b) {
                    if (!(v > 0)) bad = true;
                    S[a][a] = sqrt(v);
                } else {
                    S[a][b] = v / S[b][b];
                }
            }
        }
        for (int a = 0; a < c; a++) {
            double v = tvec[a];
            for (int q = 0; q < a; q++) v -= S[a][q] * tvec[q];
            tvec[a] = v / S[a][a];
        }
        for (int a = 0; a < c; a++) {
            for (int b = 0; b <= a; b++) L[(int64_t)(m + a) * ldl + m + b] = S[a][b];
            for (int b = a + 1; b < c; b++) L[(int64_t)(m + a) * ldl + m + b] = 0.0;
            alpha[m + a] = tvec[a];
        }
        if (bad) atomicOr(status, 1);  // Gram matrix not positive definite
    }
}

}  // namespace ital

extern "C" int ital_chol_append(const double* XT, const double* XTn, int ldx, double* L, int ldl, double* alpha,
                                const double* ynew, int m, int c, double var, double length_scale, double noise,
                                int* status, hipStream_t stream) {
    if (c < 1 || c > 16) return ital_fail(-22, "ital_chol_append: c must be in 1..16");
    if (m < 0 || m + c > ldl) return ital_fail(-22, "ital_chol_append: factor capacity exceeded");
    hipLaunchKernelGGL(ital::chol_append_kernel, dim3(1), dim3(64 * c), 0, stream, XT, XTn, ldx, L, ldl, alpha, ynew, m,
                       c, var, -2.0 * length_scale * length_scale, noise, status);
    return ital_check_launch("ital_chol_append");
}
