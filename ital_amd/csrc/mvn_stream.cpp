// Host bookkeeping of SciPy mvndst's internal random stream (MVNUNI, L'Ecuyer 1996 combined MRG) below the C ABI:
// where the stream stands, how many uniforms a call consumes, and the jump-ahead tables the scorers apply to reach the
// offset of a given (candidate, pattern, call) in the reference's serial order (reference ital/ital.py:380 ->
// scipy.stats.mvn.mvndst; its generator state is a Fortran SAVE variable: process-global and un-seedable).
// Pure integer / scalar host code, HOST pointers, no HIP: a host in any language drives t >= 3 with these alone.
#include <math.h>
#include <stdint.h>

#include "ital_hip.h"

int ital_fail(int code, const char* msg);   // api.hip

namespace {

typedef unsigned long long u64;
constexpr u64 M1 = 2147483647ULL, M2 = 2145483479ULL;
constexpr int SEED[6] = {15485857, 17329489, 36312197, 55911127, 75906931, 96210113};
constexpr int PRIMES[10] = {31, 47, 73, 113, 173, 263, 397, 593, 907, 1361};
// Keast's optimal Korobov generators C(NP, NDIM-1), NP = min(NDIM, 10), for NDIM = 2..19 (Genz, MVNDST)
constexpr int KOROBOV_C[20] = {0, 0, 13, 28, 27, 28, 20, 92, 102, 339, 206, 422, 134, 518, 134, 134, 518, 652, 382, 206};

struct Mat { u64 a[9]; };

// one step of each component as a matrix acting on (x_{n-3}, x_{n-2}, x_{n-1})^T
Mat step_matrix(int which) {
    Mat m = {{0, 1, 0, 0, 0, 1, 0, 0, 0}};
    if (which == 1) { m.a[6] = M1 - 183326; m.a[7] = 63308; m.a[8] = 0; }
    else { m.a[6] = M2 - 539608; m.a[7] = 0; m.a[8] = 86098; }
    return m;
}
Mat identity() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }

Mat mul(const Mat& x, const Mat& y, u64 m) {
    Mat r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            u64 s = 0;
            for (int k = 0; k < 3; k++) s = (s + x.a[3 * i + k] * y.a[3 * k + j] % m) % m;   // factors < 2^31
            r.a[3 * i + j] = s;
        }
    return r;
}

Mat power(Mat a, u64 e, u64 m) {
    Mat r = identity();
    while (e) {
        if (e & 1) r = mul(a, r, m);
        a = mul(a, a, m);
        e >>= 1;
    }
    return r;
}

void apply(const Mat& a, int* v, u64 m) {
    u64 o[3];
    for (int i = 0; i < 3; i++) {
        u64 s = 0;
        for (int k = 0; k < 3; k++) s = (s + a.a[3 * i + k] * (u64)v[k] % m) % m;
        o[i] = s;
    }
    for (int i = 0; i < 3; i++) v[i] = (int)o[i];
}

void store(long long* row, const Mat& a, const Mat& b) {
    for (int i = 0; i < 9; i++) { row[i] = (long long)a.a[i]; row[9 + i] = (long long)b.a[i]; }
}

int draws_per_call(int n) { return n <= 2 ? 0 : 8 * (2 * (n - 1) - 1); }

// VK(1) = 1/P, VK(i) = frac(C VK(i-1)), in floating point exactly as mvndst.f forms them
void korobov(int n, double* vk) {
    const int ndim = n - 1;
    const int p = PRIMES[(ndim < 10 ? ndim : 10) - 1];
    vk[0] = 1.0 / p;
    const double c = ndim >= 2 ? (double)KOROBOV_C[ndim] : 0.0;
    for (int i = 1; i < ndim; i++) vk[i] = fmod(c * vk[i - 1], 1.0);
}

}  // namespace

extern "C" int ital_mvn_seed(int state[6]) {
    if (!state) return ital_fail(-22, "ital_mvn_seed: null state");
    for (int i = 0; i < 6; i++) state[i] = SEED[i];
    return 0;
}

extern "C" int ital_mvn_draws_per_call(int n) { return draws_per_call(n); }

extern "C" int ital_mvn_advance(int state[6], int64_t n_draws) {
    if (!state || n_draws < 0) return ital_fail(-22, "ital_mvn_advance: null state or negative count");
    if (n_draws == 0) return 0;
    apply(power(step_matrix(1), (u64)n_draws, M1), state, M1);
    apply(power(step_matrix(2), (u64)n_draws, M2), state + 3, M2);
    return 0;
}

extern "C" int ital_mvn_tables(int t, long long* jump, long long* jumppat, double* vk) {
    if (t < 3 || t > ITAL_GENERIC_MAX_DIM) return ital_fail(-22, "ital_mvn_tables: dimension outside 3..ITAL_GENERIC_MAX_DIM");
    const u64 d = (u64)draws_per_call(t);
    if (jump) {         // [ITAL_JUMP_BITS][18]: 2^b calls of dimension t
        Mat j1 = power(step_matrix(1), d, M1), j2 = power(step_matrix(2), d, M2);
        for (int b = 0; b < ITAL_JUMP_BITS; b++) {
            store(jump + 18 * b, j1, j2);
            j1 = mul(j1, j1, M1);
            j2 = mul(j2, j2, M2);
        }
    }
    if (jumppat) {      // [2^t][18]: 2r calls, r = 0 .. 2^t - 1 (the prior-probability call of sign pattern r)
        if (t > ITAL_MAX_T) return ital_fail(-22, "ital_mvn_tables: pattern table only up to ITAL_MAX_T variables");
        const Mat j1 = power(step_matrix(1), 2 * d, M1), j2 = power(step_matrix(2), 2 * d, M2);
        Mat c1 = identity(), c2 = identity();
        for (int r = 0; r < (1 << t); r++) {
            store(jumppat + 18 * (int64_t)r, c1, c2);
            c1 = mul(j1, c1, M1);
            c2 = mul(j2, c2, M2);
        }
    }
    if (vk) korobov(t, vk);
    return 0;
}

extern "C" int ital_mvn_generic_tables(int nmax, long long* jump1, double* vk_all) {
    if (nmax < 0 || nmax > ITAL_GENERIC_MAX_DIM) return ital_fail(-22, "ital_mvn_generic_tables: nmax outside 0..ITAL_GENERIC_MAX_DIM");
    if (jump1) {        // [ITAL_JUMP_BITS][18]: 2^b uniforms
        Mat j1 = step_matrix(1), j2 = step_matrix(2);
        for (int b = 0; b < ITAL_JUMP_BITS; b++) {
            store(jump1 + 18 * b, j1, j2);
            j1 = mul(j1, j1, M1);
            j2 = mul(j2, j2, M2);
        }
    }
    if (vk_all) {       // [nmax + 1][nmax]: row n = generator vector of dimension n (first n - 1 entries)
        for (int64_t i = 0; i < (int64_t)(nmax + 1) * nmax; i++) vk_all[i] = 0.0;
        for (int n = 3; n <= nmax; n++) korobov(n, vk_all + (int64_t)n * nmax);
    }
    return 0;
}

extern "C" int ital_mvn_round_seeds(int state[6], int64_t n_cand, int k, int seeds[][6]) {
    if (!state || !seeds || k < 1 || k > ITAL_MAX_T || n_cand < k)
        return ital_fail(-22, "ital_mvn_round_seeds: bad arguments");
    int64_t n_alive = n_cand;
    for (int t = 1; t <= k; t++, n_alive--) {
        for (int j = 0; j < 6; j++) seeds[t][j] = state[j];
        const int rc = ital_mvn_advance(state, n_alive * (2LL << t) * draws_per_call(t));
        if (rc) return rc;
    }
    return 0;
}
