// MCMI[min] (Guo & Greiner) candidate scorer -- reference ital/mcmi.py:101-124 `ConditionalEntropy.__call__`
// as driven by MCMI_min.fetch_unlabelled (mcmi.py:48-81), role R11 of SURVEY.md.
//
// For the batch S = (members so far, candidate i) and every label pattern r in {-1,+1}^t the reference runs a
// simulated GP update (gp.updated_prediction -> extend_inv, gp.py:295-344) and predicts mean / variance of ALL
// current candidates j; here that update is the closed form on the t x t block
//     W = (Sigma_SS + noise I)^-1,  u_j = W Sigma_Sj,
//     mu'_j(r) = mu_j + u_j . (y_r - mu_S),   s'_j = max(0, s2_j - Sigma_jS u_j)        (independent of r)
//     CE_r(i)  = sum_j q log(q + eps) + (1 - q) log(1 - q + eps),   q = ndtr(-mu'_j(r) / sqrt(s'_j))
//     score(i) = min_r CE_r(i)     (first pattern kept when it is NaN, as `cur_ce < ce` in mcmi.py:121)
// Sigma_ij over the candidate block comes from ital_cov_block (dense, FP64 MFMA): the pairwise objective is the one
// place on the path that is a real matrix product (N_c x N_c x (d + m)).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "ital_hip.h"
#include "ital_internal.h"

#ifndef ITAL_COV_LDS
#define ITAL_COV_LDS 1     // covariance blocks of at least 256 x 256: the LDS-staged kernel
#endif
#ifndef ITAL_COV_LDS_MIN_TILES
#define ITAL_COV_LDS_MIN_TILES 1024   // four waves of workgroups over the 256 CUs x 2; below (3000^2: 576 tiles) the register-tiled kernel wins
#endif
#ifndef ITAL_MCMI_HOTK
#define ITAL_MCMI_HOTK 1
#endif

namespace ital {

typedef double d4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// out[i][j] = var*exp((|a_i|^2 + |b_j|^2 - 2 a_i.b_j)/s) - sum_r Va[r][i] Vb[r][j]
// Block = 4 waves stacked along i; every wave owns a (16 MI) x (16 MJ) register tile of MFMA
// tiles (v_mfma_f64_16x16x4_f64), so one 16-wide k step costs MI + MJ operand loads (32 B per lane each, 128-B row
// segments) for 4 MI MJ MFMAs; the kernel values replace the dot products in place and the whitened part is subtracted
// into the same accumulators; operands come straight from L1/L2 (the rows of a tile are shared by the neighbouring workgroups).
struct CovArgs {
    const double *Xa, *an; int64_t na;
    const double *Xb, *bn; int64_t nb;
    int ldx;
    const double* Va; int64_t ldva;
    const double* Vb; int64_t ldvb;
    int m;
    double var, s;
    double* out; int64_t ldo;
};

#ifndef ITAL_COV_SMALL
#define ITAL_COV_SMALL 1           // blocks of fewer than ITAL_COV_SMALL_BELOW default tiles run with the small tile
#endif
#ifndef ITAL_COV_SMALL_BELOW
#define ITAL_COV_SMALL_BELOW 512
#endif
#ifndef ITAL_COV_MI
#define ITAL_COV_MI 2
#endif
#ifndef ITAL_COV_MJ
#define ITAL_COV_MJ 4
#endif
constexpr int COV_MI_D = ITAL_COV_MI, COV_MJ_D = ITAL_COV_MJ;   // register tile per wave: 16 MI rows x 16 MJ columns

// ROWSUM: instead of storing the block, a workgroup walks the column tiles blockIdx.x, blockIdx.x + gridDim.x, ... and
// keeps sum_j |Sigma_ij| of its rows; out[blockIdx.x * ldo + i] receives the partial sum of that column split (summed in
// a fixed order by rowsum_reduce_kernel: no floating-point atomics, the result does not depend on scheduling).
// COV_MI / COV_MJ: the default 32 x 64 tile per wave (128 x 64 per workgroup), or 16 x 32 (64 x 32 per workgroup) for small
// blocks -- the reference's MCMI subsample of 1000 candidates is 128 workgroups of the large tile on 256 CUs, 512 of the small.
// Every output element accumulates its features in the same order whatever the tile: the same bits.
template <bool ROWSUM, int COV_MI = COV_MI_D, int COV_MJ = COV_MJ_D>
__global__ __launch_bounds__(256) void cov_block_kernel(CovArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col = lane & 15;
    const int kg = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.y * (64 * COV_MI) + wave * (16 * COV_MI);
    if (i0 >= a.na) return;
    const double* arow[COV_MI];
    const double* brow[COV_MJ];
    bool a_ok[COV_MI], b_ok[COV_MJ];
    int64_t ia[COV_MI], jb[COV_MJ];
#pragma unroll
    for (int p = 0; p < COV_MI; p++) {
        ia[p] = i0 + 16 * p + col;
        a_ok[p] = ia[p] < a.na;
        arow[p] = a.Xa + (a_ok[p] ? ia[p] : 0) * a.ldx;
    }
    double rs[COV_MI][4];
#pragma unroll
    for (int p = 0; p < COV_MI; p++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) rs[p][reg] = 0.0;
    const int64_t ntile = (a.nb + 16 * COV_MJ - 1) / (16 * COV_MJ);
    int64_t jt = blockIdx.x;
    do {
    const int64_t j0 = jt * (16 * COV_MJ);
#pragma unroll
    for (int q = 0; q < COV_MJ; q++) {
        jb[q] = j0 + 16 * q + col;
        b_ok[q] = jb[q] < a.nb;
        brow[q] = a.Xb + (b_ok[q] ? jb[q] : 0) * a.ldx;
    }
    d4 acc[COV_MI][COV_MJ];
#pragma unroll
    for (int p = 0; p < COV_MI; p++)
#pragma unroll
        for (int q = 0; q < COV_MJ; q++) acc[p][q] = (d4){0, 0, 0, 0};
    // Register double buffer over the feature steps: the operands of step k + 1 are requested before the 32 MFMAs of step k
    // are issued, so that their fetch (L2 / MALL) overlaps the matrix work instead of preceding it.
    struct Operands { double2 a01[COV_MI], a23[COV_MI], b01[COV_MJ], b23[COV_MJ]; };
    auto fetch = [&](Operands& o, int k0) {
        const int kk = k0 + 4 * kg;
        const bool in = k0 < a.ldx;
#pragma unroll
        for (int p = 0; p < COV_MI; p++) {
            o.a01[p] = o.a23[p] = (double2){0, 0};
            if (in && a_ok[p]) {
                o.a01[p] = *reinterpret_cast<const double2*>(arow[p] + kk);
                o.a23[p] = *reinterpret_cast<const double2*>(arow[p] + kk + 2);
            }
        }
#pragma unroll
        for (int q = 0; q < COV_MJ; q++) {
            o.b01[q] = o.b23[q] = (double2){0, 0};
            if (in && b_ok[q]) {
                o.b01[q] = *reinterpret_cast<const double2*>(brow[q] + kk);
                o.b23[q] = *reinterpret_cast<const double2*>(brow[q] + kk + 2);
            }
        }
    };
    auto multiply = [&](const Operands& o) {
        // one feature quadruple at a time over all eight accumulators: consecutive MFMAs never depend on each other
#pragma unroll
        for (int p = 0; p < COV_MI; p++)
#pragma unroll
            for (int q = 0; q < COV_MJ; q++) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a01[p].x, o.b01[q].x, acc[p][q], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < COV_MI; p++)
#pragma unroll
            for (int q = 0; q < COV_MJ; q++) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a01[p].y, o.b01[q].y, acc[p][q], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < COV_MI; p++)
#pragma unroll
            for (int q = 0; q < COV_MJ; q++) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a23[p].x, o.b23[q].x, acc[p][q], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < COV_MI; p++)
#pragma unroll
            for (int q = 0; q < COV_MJ; q++) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a23[p].y, o.b23[q].y, acc[p][q], 0, 0, 0);
    };
    {
        Operands even, odd;
        fetch(even, 0);
        for (int k0 = 0; k0 < a.ldx; k0 += 32) {
            fetch(odd, k0 + 16);          // zeros past the end: the extra MFMAs of an odd step count add nothing
            multiply(even);
            fetch(even, k0 + 32);
            multiply(odd);
        }
    }
    // dot products -> kernel values in place (D layout: reg -> row (i) = kg + 4*reg, column (j) = col) ...
#pragma unroll
    for (int q = 0; q < COV_MJ; q++) {
        const double bnj = b_ok[q] ? a.bn[jb[q]] : 0.0;
#pragma unroll
        for (int p = 0; p < COV_MI; p++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int64_t i = i0 + 16 * p + kg + 4 * reg;
                const double ani = i < a.na ? a.an[i] : 0.0;
                acc[p][q][reg] = a.var * exp((ani + bnj - 2 * acc[p][q][reg]) / a.s);
            }
    }
    // ... then the whitened part is subtracted by the matrix cores into the same accumulators: acc += (-Va)^T Vb
    for (int r0 = 0; r0 < a.m; r0 += 4) {
        const int r = r0 + kg;
        const bool r_ok = r < a.m;
        double av[COV_MI], bv[COV_MJ];
#pragma unroll
        for (int p = 0; p < COV_MI; p++) av[p] = (r_ok && a_ok[p]) ? -a.Va[(int64_t)r * a.ldva + ia[p]] : 0.0;
#pragma unroll
        for (int q = 0; q < COV_MJ; q++) bv[q] = (r_ok && b_ok[q]) ? a.Vb[(int64_t)r * a.ldvb + jb[q]] : 0.0;
#pragma unroll
        for (int p = 0; p < COV_MI; p++)
#pragma unroll
            for (int q = 0; q < COV_MJ; q++)
                acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[p], bv[q], acc[p][q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < COV_MJ; q++)
#pragma unroll
        for (int p = 0; p < COV_MI; p++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int64_t i = i0 + 16 * p + kg + 4 * reg;
                if (ROWSUM) rs[p][reg] += b_ok[q] ? fabs(acc[p][q][reg]) : 0.0;
                else if (i < a.na && b_ok[q]) a.out[i * a.ldo + jb[q]] = acc[p][q][reg];
            }
    jt += gridDim.x;
    } while (ROWSUM && jt < ntile);
    if (ROWSUM) {
#pragma unroll
        for (int p = 0; p < COV_MI; p++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                double v = rs[p][reg];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
                const int64_t i = i0 + 16 * p + kg + 4 * reg;
                if (col == 0 && i < a.na) a.out[(int64_t)blockIdx.x * a.ldo + i] = v;
            }
    }
}

// The same block with the operands staged through LDS: a workgroup owns a 128 x 128 tile of the output, its four waves a
// 64 x 64 quarter each (16 MFMA tiles: 64 accumulator registers per lane), and every 16-wide feature step is fetched from
// global memory ONCE per workgroup -- 128 rows of A and of B, 16 KB each, transposed into LDS ([feature][row], padded row
// stride) -- instead of once per wave: 16 flops per byte from L2 / HBM against 5.3 of the register-tiled kernel above,
// which reads every operand four times per workgroup.  The global loads of step s + 1 are issued before the 64 MFMAs of
// step s and written to the other LDS buffer after them (register + LDS double buffer, one barrier per step).
constexpr int CB_T = 128;              // tile edge
constexpr int CB_KS = 16;              // features per stage
constexpr int CB_LD = CB_T + 4;        // padded row stride of a staged tile (doubles)

__global__ __launch_bounds__(256, 2) void cov_block_lds_kernel(CovArgs a) {
    __shared__ double lds[2][2][CB_KS][CB_LD];     // [buffer][A / B][feature][row]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kg = lane >> 4;
    const int wy = wave >> 1, wx = wave & 1;       // quarter of the tile this wave owns
    const int64_t jt = blockIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * CB_T;
    // staging role of this thread: feature pair sk, sk + 1 of the rows srow + 8 u (u = 0..3) of the tile.  Eight lanes
    // read the 128 contiguous bytes of a row's 16 features (8 cache lines per load instruction of a wave); lanes 0-31
    // hold four feature pairs x eight rows, which land in 64 distinct LDS banks when written transposed.
    const int sk = 2 * ((lane & 3) + 4 * (lane >> 5)), srow = 32 * wave + ((lane >> 2) & 7);
    // rows past the end of a block are clamped to its last row: what they produce is never stored or summed
    const double* a_base = a.Xa + i0 * a.ldx + sk;
    int a_off[4];
#pragma unroll
    for (int u = 0; u < 4; u++) a_off[u] = (int)(min((int64_t)(srow + 8 * u), a.na - 1 - i0) * a.ldx);
    {
        const int64_t j0 = jt * CB_T;
        const double* b_base = a.Xb + j0 * a.ldx + sk;
        int b_off[4];
#pragma unroll
        for (int u = 0; u < 4; u++) b_off[u] = (int)(min((int64_t)(srow + 8 * u), a.nb - 1 - j0) * a.ldx);
        d4 acc[4][4];
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[p][q] = (d4){0, 0, 0, 0};
        double2 ra[4], rb[4];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ra[u] = *reinterpret_cast<const double2*>(a_base + a_off[u] + k0);
                rb[u] = *reinterpret_cast<const double2*>(b_base + b_off[u] + k0);
            }
        };
        auto stage = [&](int buf) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                lds[buf][0][sk][srow + 8 * u] = ra[u].x;
                lds[buf][0][sk + 1][srow + 8 * u] = ra[u].y;
                lds[buf][1][sk][srow + 8 * u] = rb[u].x;
                lds[buf][1][sk + 1][srow + 8 * u] = rb[u].y;
            }
        };
        fetch(0);
        stage(0);
        __syncthreads();
        const int nstep = a.ldx / CB_KS;
        for (int s_ = 0; s_ < nstep; s_++) {
            const int buf = s_ & 1;
            if (s_ + 1 < nstep) fetch((s_ + 1) * CB_KS);
#pragma unroll
            for (int j = 0; j < 4; j++) {            // four MFMA k-slices of the 16 staged features: feature 4 kg + j
                double av[4], bv[4];
#pragma unroll
                for (int p = 0; p < 4; p++) av[p] = lds[buf][0][4 * kg + j][64 * wy + 16 * p + col];
#pragma unroll
                for (int q = 0; q < 4; q++) bv[q] = lds[buf][1][4 * kg + j][64 * wx + 16 * q + col];
#pragma unroll
                for (int p = 0; p < 4; p++)
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[p], bv[q], acc[p][q], 0, 0, 0);
            }
            if (s_ + 1 < nstep) stage(buf ^ 1);
            __syncthreads();
        }
        // dot products -> kernel values in place (D layout: reg -> row (i) = kg + 4*reg, column (j) = col); the norms of
        // rows past the end are those of the last row (clamped like the rows themselves)
        const int64_t iw = i0 + 64 * wy, jw = j0 + 64 * wx;
        double ani[4][4];
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) ani[p][reg] = a.an[min(iw + 16 * p + kg + 4 * reg, a.na - 1)];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const double bnj = a.bn[min(jw + 16 * q + col, a.nb - 1)];
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++)     // the expression of cov_block_kernel, term for term
                    acc[p][q][reg] = a.var * exp((ani[p][reg] + bnj - 2 * acc[p][q][reg]) / a.s);
        }
        // the whitened part is subtracted by the matrix cores into the same accumulators: acc += (-Va)^T Vb
        for (int r0 = 0; r0 < a.m; r0 += 4) {
            const int r = r0 + kg;
            const bool r_ok = r < a.m;
            double av[4], bv[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int64_t i = iw + 16 * p + col;
                av[p] = (r_ok && i < a.na) ? -a.Va[(int64_t)r * a.ldva + i] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t j = jw + 16 * q + col;
                bv[q] = (r_ok && j < a.nb) ? a.Vb[(int64_t)r * a.ldvb + j] : 0.0;
            }
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[p], bv[q], acc[p][q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t j = jw + 16 * q + col;
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int64_t i = iw + 16 * p + kg + 4 * reg;
                    if (i < a.na && j < a.nb) a.out[i * a.ldo + j] = acc[p][q][reg];
                }
        }
    }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s part[s][i], s ascending
__global__ __launch_bounds__(256) void rowsum_reduce_kernel(const double* part, int64_t ld, int nsplit, int64_t n, int accumulate,
                                                            double* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = accumulate ? out[i] : 0.0;
    for (int s = 0; s < nsplit; s++) v += part[(int64_t)s * ld + i];
    out[i] = v;
}

// ---------------------------------------------------------------------------------------------------------
struct McmiArgs {
    int64_t n_i, pos_offset, n_all;
    const uint8_t* alive;
    const double *mu, *s2, *S, *C;
    int64_t lds_, ldc;
    ital_batch b;
    double noise, eps;
    double* ce;
};

// q log(q + eps) + (1 - q) log(1 - q + eps) with q = Phi(z).  Phi through Hart's rational (mvn_phi, ~1e-15 relative, a
// third of the instructions of the erf/erfc library route behind scipy's ndtr) and the short logarithm of
// device_math.h; agrees with the reference's value to ~1e-15, far inside the 1e-8 test tolerance.
template <class K>
__device__ __forceinline__ double entropy_term(double z, double eps, const K& kk) {
    const double q = mvn_phi(z, kk);
    const double p = 1.0 - q;
    return q * log_pos(q + eps, kk) + p * log_pos(p + eps, kk);
}

// One workgroup per candidate i; the 256 threads stride over the candidates j.  Patterns are enumerated in
// itertools.product order (variable 0 = slowest = most significant bit); TL low bits live in registers, the
// remaining high bits are an outer loop.
template <int T>
__global__ __launch_bounds__(256) void mcmi_score_kernel(McmiArgs a) {
    constexpr int TL = T < 5 ? T : 5;
    constexpr int NL = 1 << TL;
    constexpr int NH = 1 << (T - TL);
    __shared__ double Wsh[T][T];
    __shared__ double muS[T];
    __shared__ double red[4][NL];
    __shared__ int64_t mpos[T];
    const int64_t li = blockIdx.x;
    if (li >= a.n_i) return;
    if (!a.alive[li]) return;
    const int64_t gi = a.pos_offset + li;  // position of candidate i in the candidate block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        double Sg[T][T];
        for (int p = 0; p < T - 1; p++) {
            muS[p] = a.b.bmu[p];
            mpos[p] = a.b.bgpos[p];
            for (int q = 0; q < T - 1; q++) Sg[p][q] = a.b.sig[p * a.b.kmax + q];
            const double c = a.C[(int64_t)p * a.ldc + gi];
            Sg[p][T - 1] = c;
            Sg[T - 1][p] = c;
        }
        muS[T - 1] = a.mu[gi];
        mpos[T - 1] = gi;
        Sg[T - 1][T - 1] = a.s2[gi];
        // W = (Sigma + noise I)^-1 through the Cholesky factor
        double Lc[T][T], Li[T][T];
        for (int p = 0; p < T; p++)
            for (int q = 0; q <= p; q++) {
                double v = Sg[p][q] + (p == q ? a.noise : 0.0);
                for (int r = 0; r < q; r++) v -= Lc[p][r] * Lc[q][r];
                Lc[p][q] = (p == q) ? sqrt(v) : v / Lc[q][q];
            }
        for (int q = 0; q < T; q++)
            for (int p = 0; p < T; p++) {
                if (p < q) { Li[p][q] = 0; continue; }
                double v = (p == q) ? 1.0 : 0.0;
                for (int r = q; r < p; r++) v -= Lc[p][r] * Li[r][q];
                Li[p][q] = v / Lc[p][p];
            }
        for (int p = 0; p < T; p++)
            for (int q = 0; q <= p; q++) {
                double w = 0;
                for (int r = p; r < T; r++) w += Li[r][p] * Li[r][q];
                Wsh[p][q] = w;
                Wsh[q][p] = w;
            }
    }
    __syncthreads();
    double W[T][T], ms[T];
#pragma unroll
    for (int p = 0; p < T; p++) {
        ms[p] = muS[p];
#pragma unroll
        for (int q = 0; q < T; q++) W[p][q] = Wsh[p][q];
    }
    const double* Srow = a.S + li * a.lds_;
#if ITAL_MCMI_HOTK
    HotK kk;          // exp / log coefficients as vector-register operands (device_math.h)
    kk.load();
#else
    const LitK kk;
#endif
    double best = 0.0;
    for (int hi = 0; hi < NH; hi++) {
        double acc[NL];
#pragma unroll
        for (int r = 0; r < NL; r++) acc[r] = 0.0;
        for (int64_t j = tid; j < a.n_all; j += 256) {
            bool member = false;
#pragma unroll
            for (int p = 0; p < T - 1; p++) member = member || (mpos[p] == j);
            if (member) continue;  // already picked: no longer in learner.candidates (mcmi.py:79)
            double c[T], u[T];
#pragma unroll
            for (int p = 0; p < T - 1; p++) c[p] = a.C[(int64_t)p * a.ldc + j];
            c[T - 1] = Srow[j];
            double quad = 0, base = a.mu[j];
#pragma unroll
            for (int p = 0; p < T; p++) {
                double v = 0;
#pragma unroll
                for (int q = 0; q < T; q++) v = fma(W[p][q], c[q], v);
                u[p] = v;
                quad = fma(c[p], v, quad);
                base = fma(-v, ms[p], base);
            }
            const double sv = fmax(0.0, a.s2[j] - quad);
            // high-order variables (an outer-loop constant per hi)
#pragma unroll
            for (int p = 0; p < T - TL; p++) base += ((hi >> (T - TL - 1 - p)) & 1) ? u[p] : -u[p];
            double val[NL];
            double lo = base;
#pragma unroll
            for (int p = T - TL; p < T; p++) lo -= u[p];
            val[0] = lo;
#pragma unroll
            for (int bit = 0; bit < TL; bit++) {
                const double two_u = 2.0 * u[T - 1 - bit];
#pragma unroll
                for (int r = 0; r < (1 << bit); r++) val[r | (1 << bit)] = val[r] + two_u;
            }
            if (sv > 0) {
                const double inv_sd = 1.0 / sqrt(sv);
#pragma unroll
                for (int r = 0; r < NL; r++) acc[r] += entropy_term(-val[r] * inv_sd, a.eps, kk);
            } else {
                // norm.cdf(0, mean, 0) is NaN (scipy scale check): the whole sum becomes NaN
#pragma unroll
                for (int r = 0; r < NL; r++) acc[r] = __builtin_nan("");
            }
        }
#pragma unroll
        for (int r = 0; r < NL; r++) {
            const double v = wave_sum(acc[r]);
            if (lane == 0) red[wave][r] = v;
        }
        __syncthreads();
        if (tid == 0) {
            for (int r = 0; r < NL; r++) {
                const double v = (red[0][r] + red[1][r]) + (red[2][r] + red[3][r]);
                if ((hi == 0 && r == 0) || v < best) best = v;
            }
        }
        __syncthreads();
    }
    if (tid == 0) a.ce[li] = best;
}

// ---- batches of 5 .. 8: the same objective as three phases, each with registers of its own ------------------------------
// The kernel above holds W (T^2), 32 pattern values and 32 accumulators per thread: from T = 5 on that is 256 VGPRs plus
// 64 - 184 AGPRs at ONE wave per SIMD (T = 8: 1040 B of scratch on top).  Split (as the ITAL scorer was):
//   mcmi_prep_kernel<T>        thread per candidate i: W = (Sigma_S + noise I)^-1 and the means of S -> workspace
//   mcmi_split_kernel<T, TLR>  workgroup per (candidate i, group of high-order label bits): 2^TLR patterns in registers
//                              (TLR = 3: 8 values + 8 accumulators), W read from LDS at use; the last workgroup of a
//                              candidate (ticket counter) takes the minimum over the groups' partial minima
// The u = W c products are recomputed by every group (T^2 + 2T FMAs per j against 2^TLR entropy terms of ~150
// instructions: 11 % at T = 8, TLR = 3).  Sums over j are formed by the same threads in the same order as above (values
// agree with the single kernel's to the last bits).  Measured, 1000 candidates: t = 6 0.42 -> 0.36 ms, 7 0.80 -> 0.68, 8 1.61 -> 1.31.
#ifndef ITAL_MCMI_TLR
#define ITAL_MCMI_TLR 3
#endif

template <int T>
struct McmiSplit {
    static constexpr int TLR = ITAL_MCMI_TLR < T ? ITAL_MCMI_TLR : T;
    static constexpr int NL = 1 << TLR, NHB = 1 << (T - TLR);
    static constexpr int WDOUBLES = T * T + T;                  // W, then the means of S
    static constexpr int64_t CAND_DOUBLES = WDOUBLES + NHB + 1;  // + partial minima + ticket counter
};

template <int T>
__global__ __launch_bounds__(128) void mcmi_prep_kernel(McmiArgs a, double* __restrict__ wbuf) {
    using M = McmiSplit<T>;
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= a.n_i || !a.alive[li]) return;
    const int64_t gi = a.pos_offset + li;
    double* out = wbuf + li * M::WDOUBLES;
    double Sg[T][T];
#pragma unroll
    for (int p = 0; p < T - 1; p++) {
        out[T * T + p] = a.b.bmu[p];
#pragma unroll
        for (int q = 0; q < T - 1; q++) Sg[p][q] = a.b.sig[p * a.b.kmax + q];
        const double c = a.C[(int64_t)p * a.ldc + gi];
        Sg[p][T - 1] = c;
        Sg[T - 1][p] = c;
    }
    out[T * T + T - 1] = a.mu[gi];
    Sg[T - 1][T - 1] = a.s2[gi];
    // W = (Sigma + noise I)^-1 through the Cholesky factor (the arithmetic of mcmi_score_kernel, operation for operation)
    double Lc[T][T], Li[T][T];
#pragma unroll
    for (int p = 0; p < T; p++)
#pragma unroll
        for (int q = 0; q <= p; q++) {
            double v = Sg[p][q] + (p == q ? a.noise : 0.0);
#pragma unroll
            for (int r = 0; r < q; r++) v -= Lc[p][r] * Lc[q][r];
            Lc[p][q] = (p == q) ? sqrt(v) : v / Lc[q][q];
        }
#pragma unroll
    for (int q = 0; q < T; q++)
#pragma unroll
        for (int p = 0; p < T; p++) {
            if (p < q) { Li[p][q] = 0; continue; }
            double v = (p == q) ? 1.0 : 0.0;
#pragma unroll
            for (int r = q; r < p; r++) v -= Lc[p][r] * Li[r][q];
            Li[p][q] = v / Lc[p][p];
        }
#pragma unroll
    for (int p = 0; p < T; p++)
#pragma unroll
        for (int q = 0; q <= p; q++) {
            double w = 0;
#pragma unroll
            for (int r = p; r < T; r++) w += Li[r][p] * Li[r][q];
            out[p * T + q] = w;
            out[q * T + p] = w;
        }
}

template <int T>
__global__ __launch_bounds__(256) void mcmi_split_kernel(McmiArgs a, const double* __restrict__ wbuf, double* __restrict__ parts,
                                                         unsigned int* __restrict__ tickets) {
    using M = McmiSplit<T>;
    constexpr int TLR = M::TLR, NL = M::NL, NHB = M::NHB;
    __shared__ double Wsh[T * T + T];
    __shared__ double red[4][NL];
    __shared__ int64_t mpos[T];
    const int64_t li = blockIdx.x;
    const int hi = blockIdx.y;
    if (li >= a.n_i) return;
    if (!a.alive[li]) return;
    const int64_t gi = a.pos_offset + li;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < T * T + T) Wsh[tid] = wbuf[li * M::WDOUBLES + tid];
    if (tid < T) mpos[tid] = tid < T - 1 ? a.b.bgpos[tid] : gi;
    __syncthreads();
    double ms[T];
#pragma unroll
    for (int p = 0; p < T; p++) ms[p] = Wsh[T * T + p];
    const double* Srow = a.S + li * a.lds_;
#if ITAL_MCMI_HOTK
    HotK kk;
    kk.load();
#else
    const LitK kk;
#endif
    double acc[NL];
#pragma unroll
    for (int r = 0; r < NL; r++) acc[r] = 0.0;
    for (int64_t j = tid; j < a.n_all; j += 256) {
        bool member = false;
#pragma unroll
        for (int p = 0; p < T - 1; p++) member = member || (mpos[p] == j);
        if (member) continue;  // already picked: no longer in learner.candidates (mcmi.py:79)
        double c[T], u[T];
#pragma unroll
        for (int p = 0; p < T - 1; p++) c[p] = a.C[(int64_t)p * a.ldc + j];
        c[T - 1] = Srow[j];
        double quad = 0, base = a.mu[j];
#pragma unroll
        for (int p = 0; p < T; p++) {
            __asm__ volatile("" ::: "memory");   // W is read from LDS row by row at use: hoisted out of the j loop its T^2
                                                 // values would occupy 2 T^2 registers (what the split is there to avoid)
            double v = 0;
#pragma unroll
            for (int q = 0; q < T; q++) v = fma(Wsh[p * T + q], c[q], v);
            u[p] = v;
            quad = fma(c[p], v, quad);
            base = fma(-v, ms[p], base);
        }
        const double sv = fmax(0.0, a.s2[j] - quad);
        // high-order variables: this workgroup's share of the label patterns
#pragma unroll
        for (int p = 0; p < T - TLR; p++) base += ((hi >> (T - TLR - 1 - p)) & 1) ? u[p] : -u[p];
        double val[NL];
        double lo = base;
#pragma unroll
        for (int p = T - TLR; p < T; p++) lo -= u[p];
        val[0] = lo;
#pragma unroll
        for (int bit = 0; bit < TLR; bit++) {
            const double two_u = 2.0 * u[T - 1 - bit];
#pragma unroll
            for (int r = 0; r < (1 << bit); r++) val[r | (1 << bit)] = val[r] + two_u;
        }
        if (sv > 0) {
            const double inv_sd = 1.0 / sqrt(sv);
#pragma unroll
            for (int r = 0; r < NL; r++) acc[r] += entropy_term(-val[r] * inv_sd, a.eps, kk);
        } else {
            // norm.cdf(0, mean, 0) is NaN (scipy scale check): the whole sum becomes NaN
#pragma unroll
            for (int r = 0; r < NL; r++) acc[r] = __builtin_nan("");
        }
    }
#pragma unroll
    for (int r = 0; r < NL; r++) {
        const double v = wave_sum(acc[r]);
        if (lane == 0) red[wave][r] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double best = 0.0;
        for (int r = 0; r < NL; r++) {
            const double v = (red[0][r] + red[1][r]) + (red[2][r] + red[3][r]);
            if (r == 0 || v < best) best = v;
        }
        volatile double* pw = parts + li * NHB;
        pw[hi] = best;
        __threadfence();
        const int last = atomicAdd(tickets + li, 1u) == (unsigned)(NHB - 1);
        if (last) {
            __threadfence();
            // minimum over the groups in pattern order (`cur < ce`, mcmi.py:121-122: a NaN first value stays)
            double ce = pw[0];
            for (int h = 1; h < NHB; h++) {
                const double v = pw[h];
                if (v < ce) ce = v;
            }
            a.ce[li] = ce;
            tickets[li] = 0;
        }
    }
}

template <int T>
static int launch_mcmi(const McmiArgs& a, double* work, int64_t work_doubles, hipStream_t stream) {
    using M = McmiSplit<T>;
    if constexpr (T >= 5) {
        // batches of 5 .. 8: preparation kernel + one workgroup per (candidate, group of label patterns); the single kernel
        // of t <= 4 would need 256 registers + up to 184 accumulation registers and scratch there (it was dropped in round 4)
        if (!work) return ital_fail(-22, "ital_mcmi_score_step: t >= 5 needs the workspace of ital_mcmi_workspace(t, n_i)");
        if (work_doubles < M::CAND_DOUBLES * a.n_i)
            return ital_fail(-22, "ital_mcmi_score_step: workspace smaller than ital_mcmi_workspace(t, n_i)");
        // ticket counters first: n_i words, cleared here -- a workspace that served a call with another n_i holds that
        // call's data where this call's counters go (every finishing workgroup leaves its own counter at zero, but only its own)
        unsigned int* tickets = reinterpret_cast<unsigned int*>(work);
        if (hipMemsetAsync(tickets, 0, sizeof(unsigned int) * (size_t)a.n_i, stream) != hipSuccess)
            return ital_fail(-5, "ital_mcmi_score_step: cannot clear the ticket counters");
        double* wbuf = work + a.n_i;
        double* parts = wbuf + a.n_i * M::WDOUBLES;
        ITAL_LAUNCH(mcmi_prep_kernel<T>, dim3((unsigned)((a.n_i + 127) / 128)), dim3(128), 0, stream, a, wbuf);
        ITAL_LAUNCH((mcmi_split_kernel<T>), dim3((unsigned)a.n_i, (unsigned)M::NHB), dim3(256), 0, stream, a, wbuf, parts, tickets);
        return ital_check_launch("ital_mcmi_score_step(split)");
    } else {
        ITAL_LAUNCH(mcmi_score_kernel<T>, dim3((unsigned)a.n_i), dim3(256), 0, stream, a);
        return ital_check_launch("ital_mcmi_score_step");
    }
}

template <int T>
static int64_t mcmi_cand_doubles() { return McmiSplit<T>::CAND_DOUBLES; }

}  // namespace ital

namespace ital {

// The candidate block of a fetch (reference mcmi.py:57-66: the candidates MCMI_min scores against each other) gathered out
// of the rank's rows in ONE launch: feature rows, whitened columns, squared norms, mean and variance of the listed samples;
// a sample that lives on another rank contributes zeros (the ranks' blocks are summed afterwards).  Workgroup per
// candidate: the feature row is a coalesced copy, the m whitened values are a strided read of V's column.
__global__ __launch_bounds__(128) void gather_block_kernel(const int64_t* __restrict__ cand, int64_t nc, int64_t row0,
                                                           int64_t n_rows, const double* __restrict__ X,
                                                           const double* __restrict__ xnorm, int ldx,
                                                           const double* __restrict__ V, int64_t ldv, int m,
                                                           const double* __restrict__ mu, const double* __restrict__ s2,
                                                           double* __restrict__ Xc, double* __restrict__ Vc, int64_t ldc,
                                                           double* __restrict__ xnc, double* __restrict__ muc,
                                                           double* __restrict__ s2c) {
    const int64_t j = blockIdx.x;
    if (j >= nc) return;
    const int64_t loc = cand[j] - row0;
    const bool own = loc >= 0 && loc < n_rows;
    const double* src = X + (own ? loc : 0) * (int64_t)ldx;
    double* dst = Xc + j * (int64_t)ldx;
    for (int q = threadIdx.x; q < ldx; q += blockDim.x) dst[q] = own ? src[q] : 0.0;
    for (int r = threadIdx.x; r < m; r += blockDim.x) Vc[(int64_t)r * ldc + j] = own ? V[(int64_t)r * ldv + loc] : 0.0;
    if (threadIdx.x == 0) {
        xnc[j] = own ? xnorm[loc] : 0.0;
        muc[j] = own ? mu[loc] : 0.0;
        s2c[j] = own ? s2[loc] : 0.0;
    }
}

}  // namespace ital

using namespace ital;

extern "C" int ital_gather_block(const int64_t* cand, int64_t nc, int64_t row0, int64_t n_rows, const double* X,
                                 const double* xnorm, int ldx, const double* V, int64_t ldv, int m, const double* mu,
                                 const double* s2, double* Xc, double* Vc, int64_t ldc, double* xnc, double* muc,
                                 double* s2c, hipStream_t stream) {
    if (nc <= 0) return 0;
    if (!cand || !X || !xnorm || !mu || !s2 || !Xc || !xnc || !muc || !s2c || (m > 0 && (!V || !Vc)))
        return ital_fail(-22, "ital_gather_block: null argument");
    if (ldx <= 0 || m < 0 || ldc < nc || n_rows < 0) return ital_fail(-22, "ital_gather_block: bad sizes");
    if (nc > 0x7fffffff) return ital_fail(-22, "ital_gather_block: too many candidates per call");
    ITAL_LAUNCH(gather_block_kernel, dim3((unsigned)nc), dim3(128), 0, stream, cand, nc, row0, n_rows, X, xnorm, ldx, V, ldv, m,
                mu, s2, Xc, Vc, ldc, xnc, muc, s2c);
    return ital_check_launch("ital_gather_block");
}

extern "C" int ital_cov_block(const double* Xa, const double* an, int64_t na, const double* Xb, const double* bn,
                              int64_t nb, int ldx, const double* Va, int64_t ldva, const double* Vb, int64_t ldvb, int m,
                              double var, double length_scale, double* out, int64_t ldo, hipStream_t stream) {
    if (na <= 0 || nb <= 0) return 0;
    if (ldx % 16 != 0) return ital_fail(-22, "ital_cov_block: ldx must be a multiple of 16");
    if (m < 0 || (m > 0 && (!Va || !Vb))) return ital_fail(-22, "ital_cov_block: whitened blocks missing");
    if (ldo < nb) return ital_fail(-22, "ital_cov_block: ldo smaller than nb");
    const int64_t gx = (nb + 16 * COV_MJ_D - 1) / (16 * COV_MJ_D), gy = (na + 64 * COV_MI_D - 1) / (64 * COV_MI_D);
    if (gy > 65535) return ital_fail(-22, "ital_cov_block: too many rows per call");
    CovArgs a = {Xa, an, na, Xb, bn, nb, ldx, Va, ldva, Vb, ldvb, m, var, -2.0 * length_scale * length_scale, out, ldo};
    if (ITAL_COV_LDS && ((na + CB_T - 1) / CB_T) * ((nb + CB_T - 1) / CB_T) >= ITAL_COV_LDS_MIN_TILES) {
        // large blocks: operands staged through LDS, 128 x 128 tiles
        const int64_t lx = (nb + CB_T - 1) / CB_T, ly = (na + CB_T - 1) / CB_T;
        if (ly > 65535) return ital_fail(-22, "ital_cov_block: too many rows per call");
        // (a grid dealt to the XCDs in 8 x 8-tile patches for L2 reuse was measured: 2-9 % slower than this plain sweep)
        ITAL_LAUNCH(cov_block_lds_kernel, dim3((unsigned)lx, (unsigned)ly), dim3(256), 0, stream, a);
        return ital_check_launch("ital_cov_block(lds)");
    }
    if (ITAL_COV_SMALL && gx * gy < ITAL_COV_SMALL_BELOW) {
        // too few workgroups of the default tile to fill the chip: a quarter of the tile, four times the workgroups
        const int64_t sx = (nb + 31) / 32, sy = (na + 63) / 64;
        ITAL_LAUNCH((cov_block_kernel<false, 1, 2>), dim3((unsigned)sx, (unsigned)sy), dim3(256), 0, stream, a);
        return ital_check_launch("ital_cov_block(small)");
    }
    ITAL_LAUNCH(cov_block_kernel<false>, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, stream, a);
    return ital_check_launch("ital_cov_block");
}

extern "C" int ital_cov_abs_rowsum(const double* Xa, const double* an, int64_t na, const double* Xb, const double* bn,
                                   int64_t nb, int ldx, const double* Va, int64_t ldva, const double* Vb, int64_t ldvb,
                                   int m, double var, double length_scale, double* work, int64_t work_len, int accumulate,
                                   double* out, hipStream_t stream) {
    if (na <= 0) return 0;
    if (nb <= 0) {
        if (!accumulate && hipMemsetAsync(out, 0, (size_t)na * sizeof(double), stream) != hipSuccess)
            return ital_fail(-5, "ital_cov_abs_rowsum: memset failed");
        return 0;
    }
    if (ldx % 16 != 0) return ital_fail(-22, "ital_cov_abs_rowsum: ldx must be a multiple of 16");
    if (m < 0 || (m > 0 && (!Va || !Vb))) return ital_fail(-22, "ital_cov_abs_rowsum: whitened blocks missing");
    if (!work || work_len < na) return ital_fail(-22, "ital_cov_abs_rowsum: work area smaller than na doubles");
    const int64_t ntile = (nb + 16 * COV_MJ_D - 1) / (16 * COV_MJ_D), gy = (na + 64 * COV_MI_D - 1) / (64 * COV_MI_D);
    if (gy > 65535) return ital_fail(-22, "ital_cov_abs_rowsum: too many rows per call");
    // column splits: enough workgroups to fill the 256 CUs several times over, as many as the work area holds
    int64_t nsplit = (4096 + gy - 1) / gy;
    if (nsplit > ntile) nsplit = ntile;
    if (nsplit > work_len / na) nsplit = work_len / na;
    if (nsplit > 65535) nsplit = 65535;
    CovArgs a = {Xa, an, na, Xb, bn, nb, ldx, Va, ldva, Vb, ldvb, m, var, -2.0 * length_scale * length_scale, work, na};
    // (the row sums stay on the register-tiled kernel: with the 16 running sums on top of its 64 accumulators the
    // LDS-staged one spills at two workgroups per CU)
    ITAL_LAUNCH(cov_block_kernel<true>, dim3((unsigned)nsplit, (unsigned)gy), dim3(256), 0, stream, a);
    ITAL_LAUNCH(rowsum_reduce_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, stream, work, na, (int)nsplit, na,
                       accumulate, out);
    return ital_check_launch("ital_cov_abs_rowsum");
}

extern "C" int64_t ital_mcmi_workspace(int t, int64_t n_i) {
    if (n_i <= 0) return 0;
    switch (t) {
        case 5: return mcmi_cand_doubles<5>() * n_i;
        case 6: return mcmi_cand_doubles<6>() * n_i;
        case 7: return mcmi_cand_doubles<7>() * n_i;
        case 8: return mcmi_cand_doubles<8>() * n_i;
    }
    return 0;
}

extern "C" int ital_mcmi_score_step(const ital_mcmi_desc* d, hipStream_t stream) {
    if (!d) return ital_fail(-22, "ital_mcmi_score_step: null descriptor");
    if (d->n_i <= 0) return 0;
    if (d->t < 1 || d->t > ITAL_MAX_T) return ital_fail(-22, "ital_mcmi_score_step: batch dimension outside 1..ITAL_MAX_T");
    if (d->t > d->batch.kmax) return ital_fail(-22, "ital_mcmi_score_step: t exceeds the batch capacity");
    if (d->pos_offset < 0 || d->pos_offset + d->n_i > d->n_all)
        return ital_fail(-22, "ital_mcmi_score_step: candidate slice outside the candidate block");
    if (d->ld_cov < d->n_all || d->ldc < d->n_all) return ital_fail(-22, "ital_mcmi_score_step: leading dimension too small");
    McmiArgs a = {d->n_i, d->pos_offset, d->n_all, d->alive, d->mu, d->s2, d->cov, d->C, d->ld_cov, d->ldc, d->batch,
                  d->noise, d->eps, d->ce};
    switch (d->t) {
        case 1: return launch_mcmi<1>(a, d->work, d->work_doubles, stream);
        case 2: return launch_mcmi<2>(a, d->work, d->work_doubles, stream);
        case 3: return launch_mcmi<3>(a, d->work, d->work_doubles, stream);
        case 4: return launch_mcmi<4>(a, d->work, d->work_doubles, stream);
        case 5: return launch_mcmi<5>(a, d->work, d->work_doubles, stream);
        case 6: return launch_mcmi<6>(a, d->work, d->work_doubles, stream);
        case 7: return launch_mcmi<7>(a, d->work, d->work_doubles, stream);
        case 8: return launch_mcmi<8>(a, d->work, d->work_doubles, stream);
    }
    return ital_fail(-22, "ital_mcmi_score_step: unsupported batch dimension");
}
