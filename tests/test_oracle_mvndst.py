"""The oracle's MVNDST restatement against vectors captured from scipy.stats._mvn.mvndst (fresh process,
tests/golden/make_golden.py): values, error estimates, inform flags AND the internal random stream."""
import os

import numpy as np

from oracle import mvn


def test_mvndst_stream_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, "mvndst_stream.npz"))
    mvn.rng_reset()
    for j in range(len(z["n"])):
        n = int(z["n"][j])
        nc = n * (n - 1) // 2
        e, v, i = mvn.mvndst(z["lower"][j, :n], z["lower"][j, :n], z["infin"][j, :n], z["correl"][j, :nc],
                             maxpts=100 * n, abseps=1e-4, releps=1e-4)
        assert v == z["val"][j], (j, n)
        assert e == z["err"][j], (j, n)
        assert i == z["inform"][j], (j, n)


def test_mvndst_stream_high_dimension(golden_dir):
    """n = 13..18 incl. singular correlation matrices (duplicated variables): Korobov table beyond NDIM = 11, COVSRT's
    zero-diagonal branch and MVNDFN's limit intersection."""
    z = np.load(os.path.join(golden_dir, "mvndst_stream_hi.npz"))
    mvn.rng_reset()
    for j in range(len(z["n"])):
        n = int(z["n"][j])
        nc = n * (n - 1) // 2
        e, v, i = mvn.mvndst(z["lower"][j, :n], z["lower"][j, :n], z["infin"][j, :n], z["correl"][j, :nc],
                             maxpts=100 * n, abseps=1e-4, releps=1e-4)
        assert v == z["val"][j], (j, n, v, z["val"][j])
        assert e == z["err"][j], (j, n)
        assert i == z["inform"][j], (j, n)


def test_draw_count_per_call(golden_dir):
    z = np.load(os.path.join(golden_dir, "mvndst_stream.npz"))
    mvn.rng_reset()
    for j in range(32):
        n = int(z["n"][j])
        nc = n * (n - 1) // 2
        before = mvn.rng_draws()
        mvn.mvndst(z["lower"][j, :n], z["lower"][j, :n], z["infin"][j, :n], z["correl"][j, :nc], maxpts=100 * n,
                   abseps=1e-4, releps=1e-4)
        assert mvn.rng_draws() - before == mvn.draws_per_call(n)


def test_phi_known_answers(golden_dir):
    z = np.load(os.path.join(golden_dir, "mvndst_stream.npz"))
    got = np.array([mvn.phi(float(x)) for x in z["phi_z"]])
    assert np.array_equal(got, z["phi_val"])


def test_bivariate_known_answers(golden_dir):
    z = np.load(os.path.join(golden_dir, "mvndst_stream.npz"))
    for a0, a1, r, i0, i1, val in z["bvn"]:
        _, v, _ = mvn.mvndst([a0, a1], [a0, a1], [int(i0), int(i1)], [r], maxpts=200, abseps=1e-4, releps=1e-4)
        assert abs(v - val) <= 4e-16, (a0, a1, r, v, val)


def test_phinv_inverts_phi():
    for p in np.concatenate([np.linspace(1e-12, 1 - 1e-12, 101), 10.0 ** -np.arange(3, 300, 17)]):
        x = mvn.phinv(float(p))
        if 1e-300 < p < 1 - 1e-10:
            assert abs(mvn.phi(x) - p) <= 1e-14 * max(p, 1e-300) + 2e-16
