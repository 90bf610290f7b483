// Pieces shared by the lattice-rule scorers (score.hip: perfect-user fast path; score_generic.hip: every other
// user model / the change-estimation subset): Genz's lattice sizes, packed-triangle indexing, wave-uniform values and
// the wave-compacted Phi^-1.
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.h"

namespace ital {

constexpr int P_TAB[10] = {31, 47, 73, 113, 173, 263, 397, 593, 907, 1361};

__device__ __forceinline__ int pidx(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower, 0-based, j <= i

__device__ __forceinline__ double uniform_f64(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Four independent Phi^-1 arguments per lane.  Every lane runs the cheap central branch of AS241; the ~15 % of the
// arguments that fall into the tails (|p - 1/2| > 0.425) are compacted across the wave through LDS so that the
// expensive log/sqrt branch runs on full waves of tail arguments only (typically once per 256 inversions instead of
// once per 64).
// q: wave-private LDS queue of 64*NC doubles.
template <int NC>
__device__ __forceinline__ void phinv_wave(const double (&p)[NC], double (&out)[NC], double* __restrict__ q, int lane) {
    bool need[NC];
    int slot[NC];
    int total = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        need[c] = !phinv_is_central(p[c]);
        out[c] = phinv_central(p[c]);   // garbage (never a trap) for tail arguments: replaced below
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
        if (sl < total) q[sl] = phinv_tail(q[sl]);   // in place: each lane rewrites the slot it read
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int c = 0; c < NC; c++)
        if (need[c]) out[c] = q[slot[c]];
}

__device__ __forceinline__ void phinv_wave4(const double (&p)[4], double (&out)[4], double* __restrict__ q, int lane) {
    phinv_wave<4>(p, out, q, lane);
}

}  // namespace ital
