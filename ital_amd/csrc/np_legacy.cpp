// numpy's legacy global generator, host side: the stream the reference's Monte-Carlo pattern sampler draws from
// (reference ital/ital.py:297: scipy.stats.multivariate_normal.rvs -> np.random's RandomState.standard_normal, one
// t*mc x t block of standard normals per live candidate, in candidate-list order).
//
// The generator cannot jump: a standard normal is made by the polar method from MT19937 doubles with a data-dependent
// number of rejections (numpy/random/src/legacy/legacy-distributions.c `legacy_gauss`, numpy/random/src/mt19937), so a
// rank that scores only its own candidates still has to walk over everybody else's normals.  What it does not have to do
// is COMPUTE them: skipping needs the raw draws and the `r2 < 1` accept test only -- no log, no sqrt, no stores -- and
// runs ~10x faster than numpy produces them.  Pure host code, no HIP: usable (and tested) without a GPU.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#include "ital_hip.h"

int ital_fail(int code, const char* msg);   // api.hip

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfU, UPPER_MASK = 0x80000000U, LOWER_MASK = 0x7fffffffU;

struct Walker {
    uint32_t key[MT_N];  // state vector
    int pos;             // next word of the block
    uint32_t out[MT_N];  // tempered words of the current block

    void temper_block() {
        for (int i = 0; i < MT_N; i++) {
            uint32_t y = key[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680U;
            y ^= (y << 15) & 0xefc60000U;
            y ^= (y >> 18);
            out[i] = y;
        }
    }
    void regen() {
        uint32_t* mt = key;
        int kk = 0;
        for (; kk < MT_N - MT_M; kk++) {
            const uint32_t y = (mt[kk] & UPPER_MASK) | (mt[kk + 1] & LOWER_MASK);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
        }
        for (; kk < MT_N - 1; kk++) {
            const uint32_t y = (mt[kk] & UPPER_MASK) | (mt[kk + 1] & LOWER_MASK);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
        }
        const uint32_t y = (mt[MT_N - 1] & UPPER_MASK) | (mt[0] & LOWER_MASK);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
        temper_block();
        pos = 0;
    }
    inline uint32_t next() {
        if (pos == MT_N) regen();
        return out[pos++];
    }
    // mt19937_next_double: 53 random bits from two words
    inline double next_double() {
        const int32_t a = (int32_t)(next() >> 5), b = (int32_t)(next() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // one accepted point of the polar method (x1, x2, r2 exactly as legacy_gauss forms them)
    inline void accepted(double& x1, double& x2, double& r2) {
        do {
            x1 = 2.0 * next_double() - 1.0;
            x2 = 2.0 * next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
    }
    // `pairs` accepted points -> 2 * pairs standard normals (legacy_gauss returns f*x2 first and keeps f*x1 for the next call)
    void fill_pairs(double* out, int64_t pairs) {
        for (int64_t i = 0; i < pairs; i++) {
            double x1, x2, r2;
            accepted(x1, x2, r2);
            const double f = sqrt(-2.0 * log(r2) / r2);
            out[2 * i] = f * x2;
            out[2 * i + 1] = f * x1;
        }
    }
    // the same walk over `pairs` accepted points without keeping them; whole attempts are taken from the block while four
    // words are left in it
    void skip_pairs(int64_t pairs) {
        while (pairs > 0) {
            if (pos + 4 <= MT_N) {
                const uint32_t* w = out + pos;
                const int32_t a1 = (int32_t)(w[0] >> 5), b1 = (int32_t)(w[1] >> 6);
                const int32_t a2 = (int32_t)(w[2] >> 5), b2 = (int32_t)(w[3] >> 6);
                pos += 4;
                const double x1 = 2.0 * ((a1 * 67108864.0 + b1) / 9007199254740992.0) - 1.0;
                const double x2 = 2.0 * ((a2 * 67108864.0 + b2) / 9007199254740992.0) - 1.0;
                const double r2 = x1 * x1 + x2 * x2;
                pairs -= !(r2 >= 1.0 || r2 == 0.0);
            } else {
                double x1, x2, r2;
                accepted(x1, x2, r2);
                pairs--;
            }
        }
    }
};

}  // namespace

extern "C" int ital_np_legacy_normals(ital_np_legacy_state* st, int64_t n_skip, double* out, int64_t n_out, int threads) {
    if (!st || n_skip < 0 || n_out < 0 || (n_out > 0 && !out)) return ital_fail(-22, "ital_np_legacy_normals: bad arguments");
    if (st->pos < 0 || st->pos > MT_N) return ital_fail(-22, "ital_np_legacy_normals: state position outside 0..624");
    Walker w;
    memcpy(w.key, st->key, sizeof(w.key));
    w.pos = st->pos;
    w.temper_block();
    bool has = st->has_gauss != 0;
    double cached = st->gauss;
    // ---- skip
    if (n_skip > 0 && has) {
        has = false;
        cached = 0.0;
        n_skip--;
    }
    w.skip_pairs(n_skip / 2);
    if (n_skip & 1) {          // an odd count ends in the middle of a pair: its second value is what the next draw returns
        double x1, x2, r2;
        w.accepted(x1, x2, r2);
        const double f = sqrt(-2.0 * log(r2) / r2);
        cached = f * x1;
        has = true;
    }
    // ---- fill
    int64_t i = 0;
    if (n_out > 0 && has) {
        out[i++] = cached;
        has = false;
        cached = 0.0;
    }
    const int64_t pairs = (n_out - i) / 2;
    if (threads > 1 && pairs >= (int64_t)threads * 4096) {
        // the walk is serial, the logarithms are not: one cheap skipping pass leaves a copy of the generator at the start
        // of every thread's share, the threads then produce their shares side by side (same values, same order)
        const int64_t share = (pairs + threads - 1) / threads;
        std::vector<Walker> starts;
        std::vector<std::thread> pool;
        for (int64_t p0 = 0; p0 < pairs; p0 += share) {
            starts.push_back(w);
            w.skip_pairs(p0 + share <= pairs ? share : pairs - p0);
        }
        for (size_t c = 0; c < starts.size(); c++) {
            const int64_t p0 = (int64_t)c * share;
            const int64_t np = p0 + share <= pairs ? share : pairs - p0;
            pool.emplace_back([&starts, c, out, i, p0, np]() { starts[c].fill_pairs(out + i + 2 * p0, np); });
        }
        for (auto& t : pool) t.join();
    } else {
        w.fill_pairs(out + i, pairs);
    }
    i += 2 * pairs;
    if (i < n_out) {
        double x1, x2, r2;
        w.accepted(x1, x2, r2);
        const double f = sqrt(-2.0 * log(r2) / r2);
        out[i] = f * x2;
        cached = f * x1;
        has = true;
    }
    memcpy(st->key, w.key, sizeof(w.key));
    st->pos = w.pos;
    st->has_gauss = has ? 1 : 0;
    st->gauss = cached;
    return 0;
}
