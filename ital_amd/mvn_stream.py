"""Host bookkeeping of SciPy mvndst's internal random stream (MVNUNI, L'Ecuyer 1996 combined MRG).

The reference draws its lattice shifts from a process-global, un-seedable generator inside
scipy.stats.mvn.mvndst (reference ital/ital.py:380).  The device scorer replays that
stream: this module knows where the stream stands (one global position per process, like the Fortran SAVE
state), how many uniforms a call of dimension n consumes, and builds the jump-ahead matrices the kernel
applies to reach the offset of a given (candidate, pattern, call) in the reference's serial order.
Pure integer arithmetic, no GPU needed.
"""
import numpy as np

M1, M2 = 2147483647, 2145483479
SEED = (15485857, 17329489, 36312197, 55911127, 75906931, 96210113)
# one step of each component as a matrix acting on (x_{n-3}, x_{n-2}, x_{n-1})^T
A1 = ((0, 1, 0), (0, 0, 1), ((-183326) % M1, 63308, 0))
A2 = ((0, 1, 0), (0, 0, 1), ((-539608) % M2, 0, 86098))

PRIMES = (31, 47, 73, 113, 173, 263, 397, 593, 907, 1361)
# Keast's optimal Korobov generators C(NP, NDIM-1), NP = min(NDIM, 10), for NDIM = 2..19 (Genz, MVNDST; pinned against
# SciPy by tests/golden/mvndst_stream.npz and mvndst_stream_hi.npz through the oracle's restatement)
KOROBOV_C = {2: 13, 3: 28, 4: 27, 5: 28, 6: 20, 7: 92, 8: 102, 9: 339, 10: 206, 11: 422, 12: 134, 13: 518, 14: 134,
             15: 134, 16: 518, 17: 652, 18: 382, 19: 206}


def _matmul(a, b, m):
    return tuple(tuple(sum(a[i][k] * b[k][j] for k in range(3)) % m for j in range(3)) for i in range(3))


def _matpow(a, e, m):
    r = ((1, 0, 0), (0, 1, 0), (0, 0, 1))
    while e:
        if e & 1:
            r = _matmul(a, r, m)
        a = _matmul(a, a, m)
        e >>= 1
    return r


def _apply(mat, vec, m):
    return tuple(sum(mat[i][k] * vec[k] for k in range(3)) % m for i in range(3))


def draws_per_call(n):
    """Uniforms consumed by one mvndst call with n finite-limit variables (8 shifts x (NDIM-1 shuffle draws +
    NDIM shifts), NDIM = n-1); the closed forms n <= 2 draw nothing."""
    return 0 if n <= 2 else 8 * (2 * (n - 1) - 1)


def korobov_vk(n):
    """Generator vector of the lattice rule used for n variables: VK(1) = 1/P, VK(i) = frac(C * VK(i-1)),
    evaluated in floating point exactly as SciPy's mvndst.f does."""
    ndim = n - 1
    p = PRIMES[min(ndim, 10) - 1]
    vk = np.empty(ndim, dtype=np.float64)
    vk[0] = 1.0 / p
    c = float(KOROBOV_C[ndim]) if ndim >= 2 else 0.0
    for i in range(1, ndim):
        vk[i] = np.fmod(c * vk[i - 1], 1.0)
    return vk


_POW2 = {1: [A1], 2: [A2]}   # A^(2^b) of both components, extended on demand


def _jump_state(vec, n, which, a, m):
    """vec advanced by n steps of component `which`: one matrix-vector product per set bit of n (the squarings are
    cached), instead of a fresh square-and-multiply of 3x3 matrices per call."""
    pows = _POW2[which]
    b = 0
    while n:
        if b >= len(pows):
            pows.append(_matmul(pows[-1], pows[-1], m))
        if n & 1:
            vec = _apply(pows[b], vec, m)
        n >>= 1
        b += 1
    return vec


class MvnStream:
    """Position of the MVNUNI stream; `state` is the generator state after `draws` uniforms."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.state = SEED
        self.draws = 0

    def advance(self, n):
        n = int(n)
        if n <= 0:
            return
        self.state = _jump_state(self.state[:3], n, 1, A1, M1) + _jump_state(self.state[3:], n, 2, A2, M2)
        self.draws += n

    def peek(self, n):
        """State after n more draws, without moving."""
        n = int(n)
        return _jump_state(self.state[:3], n, 1, A1, M1) + _jump_state(self.state[3:], n, 2, A2, M2)


_jump_cache = {}


def jump_table(n, bits=48):
    """int64 [bits][18]: transition matrices of both components for 2^b calls of dimension n."""
    key = (n, bits)
    if key not in _jump_cache:
        d = draws_per_call(n)
        j1, j2 = _matpow(A1, d, M1), _matpow(A2, d, M2)
        out = np.empty((bits, 18), dtype=np.int64)
        for b in range(bits):
            out[b, :9] = np.array(j1, dtype=np.int64).ravel()
            out[b, 9:] = np.array(j2, dtype=np.int64).ravel()
            j1, j2 = _matmul(j1, j1, M1), _matmul(j2, j2, M2)
        _jump_cache[key] = out
    return _jump_cache[key]


def jump_pattern_table(n):
    """int64 [2^n][18]: transition matrices of both components for 2r calls of dimension n, r = 0..2^n-1 (the reference
    evaluates, per sign pattern r, the prior probability -- call 2r of the candidate -- and the probability after the
    simulated update, ital.py:191-206)."""
    key = ("pattern", n)
    if key not in _jump_cache:
        d = 2 * draws_per_call(n)
        j1, j2 = _matpow(A1, d, M1), _matpow(A2, d, M2)
        c1 = c2 = ((1, 0, 0), (0, 1, 0), (0, 0, 1))
        out = np.empty((1 << n, 18), dtype=np.int64)
        for r in range(1 << n):
            out[r, :9] = np.array(c1, dtype=np.int64).ravel()
            out[r, 9:] = np.array(c2, dtype=np.int64).ravel()
            c1, c2 = _matmul(j1, c1, M1), _matmul(j2, c2, M2)
        _jump_cache[key] = out
    return _jump_cache[key]


def jump1_table(bits=48):
    """int64 [bits][18]: transition matrices of both components for 2^b uniforms."""
    key = ("single", bits)
    if key not in _jump_cache:
        j1, j2 = A1, A2
        out = np.empty((bits, 18), dtype=np.int64)
        for b in range(bits):
            out[b, :9] = np.array(j1, dtype=np.int64).ravel()
            out[b, 9:] = np.array(j2, dtype=np.int64).ravel()
            j1, j2 = _matmul(j1, j1, M1), _matmul(j2, j2, M2)
        _jump_cache[key] = out
    return _jump_cache[key]


def vk_table(nmax):
    """float64 [nmax + 1][nmax]: Korobov generator vector of every dimension 3..nmax (row n, first n - 1 entries)."""
    out = np.zeros((nmax + 1, nmax), dtype=np.float64)
    for n in range(3, nmax + 1):
        out[n, : n - 1] = korobov_vk(n)
    return out


#: the process-global stream (one per process, as in the reference)
GLOBAL = MvnStream()
