// MVNUNI state at the first call of every candidate of a greedy step (jump-ahead by the candidate's rank among the live
// list positions), shared by the seed kernel of the lattice scorer (score.hip) and by the covariance-column launch that
// carries it piggyback (rbf.hip: the seeds of step t + 1 depend on nothing but the selection of step t, like the column).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"

namespace ital {

struct SeedArgs {
    const uint8_t* alive;   // [n] position still a candidate
    const int64_t* gpos;    // [n] global list position of every local position, or nullptr: pos_offset + p
    int64_t pos_offset;
    const int64_t* bgpos;   // list positions of the batch members picked so far
    int nprev;              // how many
    int seed[6];            // generator state at the first call of the step
    const long long* jump;  // [48][18]: transition matrices for 2^b calls of the step's dimension
    int ncalls;             // calls per candidate
    int64_t slab_lo, slab_n;
    int* seeds;             // [slab_n][6] out
};

__device__ __forceinline__ void qmc_seed_body(const SeedArgs& a, int64_t i) {
    if (i >= a.slab_n) return;
    const int64_t p = a.slab_lo + i;
    if (!a.alive[p]) return;
    const int64_t gpos = a.gpos ? a.gpos[p] : a.pos_offset + p;
    int64_t before = gpos;
    for (int q = 0; q < a.nprev; q++) before -= (a.bgpos[q] < gpos) ? 1 : 0;
    uint64_t calls_before = (uint64_t)before * (uint64_t)a.ncalls;
    MrgState rng = {a.seed[0], a.seed[1], a.seed[2], a.seed[3], a.seed[4], a.seed[5]};
    for (int bit = 0; calls_before != 0; bit++, calls_before >>= 1)
        if (calls_before & 1) mrg_apply(rng, a.jump + bit * 18);
    int* sp = a.seeds + i * 6;
    sp[0] = rng.x10; sp[1] = rng.x11; sp[2] = rng.x12; sp[3] = rng.x20; sp[4] = rng.x21; sp[5] = rng.x22;
}

}  // namespace ital
