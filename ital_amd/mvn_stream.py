"""Host bookkeeping of SciPy mvndst's internal random stream (MVNUNI, L'Ecuyer 1996 combined MRG).

The reference draws its lattice shifts from a process-global, un-seedable generator inside
scipy.stats.mvn.mvndst (reference ital/ital.py:380).  The device scorer replays that
stream: this module knows where the stream stands (one global position per process, like the Fortran SAVE
state), how many uniforms a call of dimension n consumes, and hands out the jump-ahead matrices the kernel
applies to reach the offset of a given (candidate, pattern, call) in the reference's serial order.

The arithmetic lives below the C ABI (csrc/mvn_stream.cpp: ital_mvn_seed / _advance / _tables / _generic_tables /
_draws_per_call -- host code of libital_hip.so, no GPU needed), so that a host in any language can drive the
scorers for t >= 3; this module is the ctypes wrapper plus the per-process position.
"""
import ctypes

import numpy as np

from . import _lib

M1, M2 = 2147483647, 2145483479
SEED = (15485857, 17329489, 36312197, 55911127, 75906931, 96210113)
PRIMES = (31, 47, 73, 113, 173, 263, 397, 593, 907, 1361)
_JUMP_BITS = _lib.ITAL_JUMP_BITS


def draws_per_call(n):
    """Uniforms consumed by one mvndst call with n finite-limit variables (8 shifts x (NDIM-1 shuffle draws +
    NDIM shifts), NDIM = n-1); the closed forms n <= 2 draw nothing."""
    return 0 if n <= 2 else 8 * (2 * (n - 1) - 1)


def _state6(state):
    return (ctypes.c_int * 6)(*[int(v) for v in state])


def _advanced(state, n):
    st = _state6(state)
    _lib.check(_lib.lib().ital_mvn_advance(st, int(n)))
    return tuple(int(v) for v in st)


def korobov_vk(n):
    """Generator vector of the lattice rule used for n variables: VK(1) = 1/P, VK(i) = frac(C * VK(i-1)),
    evaluated in floating point exactly as SciPy's mvndst.f does."""
    vk = np.empty(n - 1, dtype=np.float64)
    _lib.check(_lib.lib().ital_mvn_tables(int(n), None, None, vk.ctypes.data))
    return vk


class MvnStream:
    """Position of the MVNUNI stream; `state` is the generator state after `draws` uniforms."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.state = SEED          # = ital_mvn_seed (checked in tests/test_host_logic.py); no library call at import time
        self.draws = 0

    def advance(self, n):
        n = int(n)
        if n <= 0:
            return
        self.state = _advanced(self.state, n)
        self.draws += n

    def peek(self, n):
        """State after n more draws, without moving."""
        return _advanced(self.state, max(int(n), 0))


_cache = {}


def jump_table(n, bits=_JUMP_BITS):
    """int64 [bits][18]: transition matrices of both components for 2^b calls of dimension n."""
    key = ("calls", n)
    if key not in _cache:
        out = np.empty((_JUMP_BITS, 18), dtype=np.int64)
        _lib.check(_lib.lib().ital_mvn_tables(int(n), out.ctypes.data, None, None))
        _cache[key] = out
    return _cache[key][:bits]


def jump_pattern_table(n):
    """int64 [2^n][18]: transition matrices of both components for 2r calls of dimension n, r = 0..2^n-1 (the reference
    evaluates, per sign pattern r, the prior probability -- call 2r of the candidate -- and the probability after the
    simulated update, ital.py:191-206)."""
    key = ("pattern", n)
    if key not in _cache:
        out = np.empty((1 << n, 18), dtype=np.int64)
        _lib.check(_lib.lib().ital_mvn_tables(int(n), None, out.ctypes.data, None))
        _cache[key] = out
    return _cache[key]


def jump1_table(bits=_JUMP_BITS):
    """int64 [bits][18]: transition matrices of both components for 2^b uniforms."""
    key = "single"
    if key not in _cache:
        out = np.empty((_JUMP_BITS, 18), dtype=np.int64)
        _lib.check(_lib.lib().ital_mvn_generic_tables(0, out.ctypes.data, None))
        _cache[key] = out
    return _cache[key][:bits]


def vk_table(nmax):
    """float64 [nmax + 1][nmax]: Korobov generator vector of every dimension 3..nmax (row n, first n - 1 entries)."""
    out = np.zeros((nmax + 1, nmax), dtype=np.float64)
    _lib.check(_lib.lib().ital_mvn_generic_tables(int(nmax), None, out.ctypes.data))
    return out


#: the process-global stream (one per process, as in the reference)
GLOBAL = MvnStream()
