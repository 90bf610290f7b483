"""CPU tests of the product's host logic: the MVNUNI stream bookkeeping (jump-ahead matrices, draw counts,
Korobov generators), and the C ABI (the library loads and exports everything include/ital_hip.h declares)."""
import ctypes
import os
import re

import numpy as np

from ital_amd import mvn_stream as ms
from oracle import mvn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_advance_matches_generator():
    mvn.rng_reset()
    s = ms.MvnStream()
    for n in [1, 2, 24, 40, 999, 54321]:
        mvn.rng_skip(n)
        s.advance(n)
        assert list(s.state) == mvn.rng_state()
    assert s.draws == mvn.rng_draws()


def _apply(J, st):
    a, b = J[:9].reshape(3, 3), J[9:].reshape(3, 3)
    x = [int(sum(int(a[i, k]) * st[k] for k in range(3)) % ms.M1) for i in range(3)]
    y = [int(sum(int(b[i, k]) * st[3 + k] for k in range(3)) % ms.M2) for i in range(3)]
    return x + y


def test_jump_table_bits():
    for n in (3, 4, 5, 8):
        jt = ms.jump_table(n)
        assert jt.shape == (48, 18) and jt.dtype == np.int64
        calls = 0b1011001
        st = list(ms.SEED)
        for b in range(8):
            if (calls >> b) & 1:
                st = _apply(jt[b], st)
        mvn.rng_reset()
        mvn.rng_skip(calls * ms.draws_per_call(n))
        assert st == mvn.rng_state()


def test_draws_per_call_matches_oracle():
    rng = np.random.default_rng(0)
    for n in range(1, 10):
        a = rng.normal(size=n)
        c = np.zeros(n * (n - 1) // 2)
        before = mvn.rng_draws()
        mvn.mvndst(a, a, np.ones(n, dtype=np.int32), c, maxpts=100 * n, abseps=1e-4, releps=1e-4)
        assert mvn.rng_draws() - before == ms.draws_per_call(n)


def test_korobov_generators():
    # first two components for every dimension used on the device
    for n in range(3, 10):
        vk = ms.korobov_vk(n)
        p = ms.PRIMES[min(n - 1, 10) - 1]
        assert vk[0] == 1.0 / p
        assert abs(vk[1] - (ms.KOROBOV_C[n - 1] % p) / p) < 1e-15


def test_c_abi_exports_every_declared_symbol():
    from ital_amd import _lib
    header = open(os.path.join(ROOT, "include", "ital_hip.h")).read()
    declared = set(re.findall(r"\b(ital_[a-z_0-9]+)\s*\(", header))
    declared -= {"ital_batch", "ital_score_desc"}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in _lib.load().ital_version()


def test_struct_layout_matches_header():
    from ital_amd import _lib
    # ital_batch: 3 ints (+pad) + 8 pointers ; ital_score_desc as declared
    assert ctypes.sizeof(_lib.ItalBatch) == 16 + 8 * 8
    d = _lib.ItalScoreDesc
    assert d.batch.offset % 8 == 0 and d.seed.size == 24
