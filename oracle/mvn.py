"""ctypes front-end to oracle/mvndst_oracle.c.  ORACLE = test infrastructure (see oracle/__init__.py).

`mvndst` has the call signature of scipy.stats.mvn.mvndst as the reference uses it
(reference ital/ital.py:380-381): returns (error, value, inform).
"""
import ctypes

import numpy as np

from .build import build

_lib = ctypes.CDLL(build())
_lib.mvn_phi.restype = ctypes.c_double
_lib.mvn_phi.argtypes = [ctypes.c_double]
_lib.mvn_phinv.restype = ctypes.c_double
_lib.mvn_phinv.argtypes = [ctypes.c_double]
_lib.mvn_bvu.restype = ctypes.c_double
_lib.mvn_bvu.argtypes = [ctypes.c_double] * 3
_lib.mvn_uni.restype = ctypes.c_double
_lib.mvn_rng_draws.restype = ctypes.c_uint64
_lib.mvn_rng_skip.argtypes = [ctypes.c_uint64]
_lib.mvndst.restype = ctypes.c_int
_lib.mvndst.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                               ctypes.c_void_p, ctypes.c_void_p]
_lib.mvndst_many.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                             ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                             ctypes.c_void_p]

phi = _lib.mvn_phi
phinv = _lib.mvn_phinv
bvu = _lib.mvn_bvu
uni = _lib.mvn_uni


def rng_reset():
    """Back to the Fortran DATA seeds (= a fresh process of the reference)."""
    _lib.mvn_rng_reset()


def rng_state():
    s = (ctypes.c_int32 * 6)()
    _lib.mvn_rng_get_state(s)
    return [int(v) for v in s]


def rng_set_state(state):
    s = (ctypes.c_int32 * 6)(*[int(v) for v in state])
    _lib.mvn_rng_set_state(s)


def rng_draws():
    return int(_lib.mvn_rng_draws())


def rng_skip(n):
    _lib.mvn_rng_skip(int(n))


def mvndst(lower, upper, infin, correl, maxpts=2000, abseps=1e-6, releps=1e-6):
    lower = np.ascontiguousarray(lower, dtype=np.float64)
    upper = np.ascontiguousarray(upper, dtype=np.float64)
    infin = np.ascontiguousarray(infin, dtype=np.int32)
    correl = np.ascontiguousarray(correl, dtype=np.float64)
    err = ctypes.c_double()
    val = ctypes.c_double()
    inform = _lib.mvndst(len(lower), lower.ctypes.data, upper.ctypes.data, infin.ctypes.data, correl.ctypes.data,
                         int(maxpts), float(abseps), float(releps), ctypes.byref(err), ctypes.byref(val))
    return err.value, val.value, inform


def draws_per_call(n):
    """Uniforms one mvndst call of n (finite-limit) variables takes from MVNUNI: 8 shifts x
    ((NDIM-1) shuffle draws + NDIM shift draws), NDIM = n-1; none for n <= 2 (closed forms)."""
    return 0 if n <= 2 else 8 * (2 * (n - 1) - 1)
