"""The context-style layer of the C ABI (ital_ctx_create / _fit / _update / _fetch /
_predict_stored, csrc/ctx.hip) against the golden USPS session of the real reference (tests/golden/usps500.npz, two rounds
of fetch_unlabelled(4) + update): the library owns every device buffer, the host passes host arrays only -- through ctypes
here, from C++ in tests/host_gpu_driver.cpp.  Reference: ital/retrieval_base.py:34-61, :105-126, ital/ital.py:84-134,
ital/gp.py:203-232."""
import ctypes
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ranks  # noqa: E402


def _session(lib, chk, z, comm=None):
    """Two golden rounds through the context API; returns nothing, asserts everything."""
    X = np.ascontiguousarray(z["X"], dtype=np.float64)
    n, d = X.shape
    k = int(z["k"])
    ctx = ctypes.c_void_p()
    chk(lib.ital_ctx_create(n, d, float(z["length_scale"]), float(z["var"]), float(z["noise"]), 64, 0, 1, comm, ctypes.byref(ctx)))
    try:
        chk(lib.ital_ctx_fit(ctx, X.ctypes.data, 0, None))
        picks = np.zeros(8, dtype=np.int64)
        # before any label: the reference dies with an AttributeError (gp.py:222); here a clean error
        assert lib.ital_ctx_fetch(ctx, k, picks.ctypes.data, None) == -22
        mean, var = np.empty(n), np.empty(n)
        for r in range(int(z["rounds"])):
            ind = np.ascontiguousarray(z["r%d_ind" % r], dtype=np.int64)       # samples labelled before this round's fetch
            y = np.ascontiguousarray(z["r%d_y" % r], dtype=np.float64)
            new = slice(0, len(ind)) if r == 0 else slice(len(z["r%d_ind" % (r - 1)]), len(ind))
            chk(lib.ital_ctx_update(ctx, ind[new].ctypes.data, y[new].ctypes.data, len(ind[new]), None))
            chk(lib.ital_ctx_predict_stored(ctx, mean.ctypes.data, var.ctypes.data, None))
            np.testing.assert_allclose(mean, z["r%d_rel_mean" % r], rtol=0, atol=2e-9)
            np.testing.assert_allclose(var, z["r%d_var" % r], rtol=0, atol=2e-9)
            got = lib.ital_ctx_fetch(ctx, k, picks.ctypes.data, None)
            assert got == k, lib.ital_last_error()
            assert picks[:k].tolist() == z["r%d_ret" % r].tolist()             # the reference's batch
        # feedback can be given once (retrieval_base.py:107-109)
        again = np.ascontiguousarray(z["r0_ind"][:1], dtype=np.int64)
        one = np.ones(1)
        assert lib.ital_ctx_update(ctx, again.ctypes.data, one.ctypes.data, 1, None) == -22
        assert b"Cannot change feedback" in lib.ital_last_error()
        row0 = ctypes.c_int64(-1)
        assert lib.ital_ctx_local_rows(ctx, ctypes.byref(row0)) == n and row0.value == 0
        # fit again: every label forgotten, the first round comes out as before
        chk(lib.ital_ctx_fit(ctx, X.ctypes.data, 0, None))
    finally:
        chk(lib.ital_ctx_destroy(ctx))


def test_context_api_replays_the_golden_session(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ital_amd import _lib
    lib = _lib.load()
    torch.cuda.init()
    _session(lib, _lib.check, np.load(os.path.join(golden_dir, "usps500.npz")))


def _comm_worker(rank, world, port, golden, out):
    dev, group = _ranks.join(rank, world, port, "rccl1")
    try:
        import torch.distributed as dist
        from ital_amd import _lib, sharding
        lib = _lib.load()
        comm = sharding.raw_comm(group, dev)            # the process group's own ncclComm_t (one rank)
        if comm is None:
            out[rank] = ("no raw communicator", sharding.raw_comm_reason(group, dev))
            return
        _session(lib, _lib.check, np.load(golden), comm=comm)
        torch.cuda.synchronize()
        out[rank] = ("ok", None)
    finally:
        _ranks.leave(group)


def test_context_api_through_the_exchange_path(golden_dir):
    """The same session with a communicator: every greedy step goes ital_select_local -> ncclAllGather -> ital_select_resolve
    (a one-rank RCCL group: all a one-GPU box can host)."""
    res = _ranks.spawn(_comm_worker, 1, os.path.join(golden_dir, "usps500.npz"))[0]
    assert res[0] == "ok", res
