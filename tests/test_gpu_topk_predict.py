"""Device top-k of the predictive means (reference retrieval_base.py:64-75) and gp.predict on external points
(reference gp.py:264-292, all three cov modes) through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return "cuda:0"


def _topk(dev, v, k, offset=0):
    from ital_amd import _lib
    lib = _lib.lib()
    t = torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    work = torch.empty(int(lib.ital_topk_workspace()), dtype=torch.uint8, device=dev)
    vals = torch.empty(k, dtype=torch.float64, device=dev)
    idx = torch.empty(k, dtype=torch.int64, device=dev)
    _lib.check(lib.ital_topk(t.data_ptr(), len(v), offset, k, vals.data_ptr(), idx.data_ptr(), work.data_ptr(),
                             torch.cuda.current_stream().cuda_stream))
    return vals.cpu().numpy(), idx.cpu().numpy()


def _expect(v, k):
    """np.argsort(v)[::-1][:k] with a stable sort: descending, NaN first, equal values by descending index."""
    return np.argsort(v, kind="stable")[::-1][:k]


@pytest.mark.parametrize("n,k", [(1, 1), (7, 7), (100, 10), (5000, 1000), (4096, 4096), (70001, 4096), (1_000_000, 100)])
def test_topk_matches_stable_argsort(dev, n, k):
    rng = np.random.default_rng(n + k)
    v = rng.normal(size=n)
    vals, idx = _topk(dev, v, k, offset=0)
    want = _expect(v, k)
    np.testing.assert_array_equal(idx, want)
    np.testing.assert_array_equal(vals, v[want])


def test_topk_ties_nans_signed_zeros_and_offsets(dev):
    rng = np.random.default_rng(0)
    v = np.round(rng.normal(size=3000), 1)            # many equal values around the threshold
    v[[5, 17, 2999]] = np.nan
    v[[40, 41]] = [0.0, -0.0]
    for k in (1, 2, 3, 4, 50, 700, 3000):
        vals, idx = _topk(dev, v, k, offset=1000)
        want = _expect(v, k)
        np.testing.assert_array_equal(idx - 1000, want)
        np.testing.assert_array_equal(vals, v[want])
    z = np.array([0.0, -0.0, 0.0, -0.0])                # numpy compares them equal: the index decides
    _, idx = _topk(dev, z, 4)
    assert idx.tolist() == [3, 2, 1, 0]


def test_topk_degenerate_all_equal(dev):
    v = np.full(20000, 0.25)                            # more ties at the threshold than the listed-ties buffer holds
    vals, idx = _topk(dev, v, 37)
    np.testing.assert_array_equal(idx, np.arange(19999, 19999 - 37, -1))
    v[123] = 0.5
    _, idx = _topk(dev, v, 3)
    np.testing.assert_array_equal(idx, [123, 19999, 19998])


def test_top_results_and_rel_mean_of_the_learner(dev):
    from ital_amd import ITAL
    rng = np.random.default_rng(4)
    X = rng.random((3000, 16))
    L = ITAL(X, length_scale=1.1, device=dev)
    assert L.rel_mean is None                            # reference retrieval_base.py:61
    L.update({3: 1, 77: -1, 1500: 1})
    m = L.rel_mean
    for k in (1, 10, 250):
        np.testing.assert_array_equal(L.top_results(k), np.argsort(m, kind="stable")[::-1][:k])
    full = L.top_results()
    assert len(full) == 3000 and full[0] == L.top_results(1)[0]
    np.testing.assert_array_equal(L.top_results(5000), full)   # k beyond the data: everything (numpy slicing semantics)


@pytest.mark.parametrize("m_lab", [1, 16, 17, 50])
def test_predict_all_cov_modes_vs_oracle(dev, m_lab):
    from ital_amd import GaussianProcess
    from oracle.gp import OracleGP
    rng = np.random.default_rng(m_lab)
    X = rng.random((400, 21))
    ls = 0.9 * np.sqrt(21 / 12.0)
    gp = GaussianProcess(X, ls, var=1.2, device=dev)
    ref = OracleGP(X, ls, var=1.2)
    idx = rng.permutation(400)[:m_lab].tolist()
    y = np.where(rng.random(m_lab) > 0.5, 1.0, -1.0)
    gp.update(idx, y)
    ref.update(idx, y)
    Xt = rng.random((37, 21))
    Xt[5] = X[idx[0]]                                    # a test point that coincides with a labelled one: variance ~ 0, clamped
    want_mean, want_cov = ref.predict(Xt, cov_mode="full")
    np.testing.assert_allclose(gp.predict(Xt), want_mean, rtol=0, atol=2e-9)
    pm, pv = gp.predict(Xt, cov_mode="diag")
    np.testing.assert_allclose(pm, want_mean, rtol=0, atol=2e-9)
    np.testing.assert_allclose(pv, np.maximum(0, np.diag(want_cov)), rtol=0, atol=2e-9)
    assert (pv >= 0).all()
    fm, fc = gp.predict(Xt, cov_mode="full")
    np.testing.assert_allclose(fm, want_mean, rtol=0, atol=2e-9)
    np.testing.assert_allclose(fc, want_cov, rtol=0, atol=2e-9)
    with pytest.raises(ValueError):
        gp.predict(Xt, cov_mode="half")
