"""One full fetch_unlabelled(4) of the reference on USPS-test 500 x 256, driven through the C ABI ALONE: nothing of
ital_amd's learners / GP / stream bookkeeping is imported -- only the ctypes signatures of include/ital_hip.h (ital_amd._lib)
and torch as the owner of device memory.  This is the sequence a host that is not Python runs;
for greedy steps t >= 3 it needs the stream tables and the stream position from the library itself
(ital_mvn_seed / ital_mvn_tables / ital_mvn_advance / ital_mvn_draws_per_call).

Reference: ital/ital.py:84-134 (fetch_unlabelled), :380 (mvndst), ital/gp.py:164-200 (update), :203-232 (predict_stored).
Expected picks: tests/golden/usps500.npz, produced by the real reference (tests/golden/make_golden.py)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))


def test_fetch_unlabelled_through_ctypes_only():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ital_amd import _lib
    lib = _lib.load()
    chk = _lib.check
    z = np.load(os.path.join(HERE, "golden", "usps500.npz"))
    X, ls, var, noise, q, k = z["X"], float(z["length_scale"]), float(z["var"]), float(z["noise"]), int(z["query"]), int(z["k"])
    n, d = X.shape
    dev = torch.device("cuda:0")
    f64, i64, i32 = torch.float64, torch.int64, torch.int32
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731

    # ---- data + GP state (what ital_amd.gp holds): features padded to 16, squared norms, Cholesky-whitened block
    ldx, ldv, cap, kmax = (d + 15) // 16 * 16, (n + 15) // 16 * 16, 16, 4
    Xd = torch.zeros((n, ldx), dtype=f64, device=dev)
    Xd[:, :d] = torch.from_numpy(X).to(dev)
    xn = torch.empty(n, dtype=f64, device=dev)
    chk(lib.ital_row_norms(P(Xd), n, ldx, P(xn), st))
    Lc = torch.zeros((cap, cap), dtype=f64, device=dev)
    alpha = torch.zeros(cap, dtype=f64, device=dev)
    XT = torch.zeros((cap, ldx), dtype=f64, device=dev)
    XTn = torch.zeros(cap, dtype=f64, device=dev)
    V = torch.zeros((cap, ldv), dtype=f64, device=dev)
    mu = torch.zeros(n, dtype=f64, device=dev)
    s2 = torch.full((n,), var, dtype=f64, device=dev)
    status = torch.zeros(1, dtype=i32, device=dev)
    # update({q: +1}) (reference gp.py:164-200): stage the row, rank-1 Cholesky append, whitened row + mean / variance refresh
    lb = _lib.ItalLabelBatch()
    lb.c, lb.slot[0], lb.y[0] = 1, q, 1.0
    ybuf = torch.empty(16, dtype=f64, device=dev)
    chk(lib.ital_stage_labelled(P(Xd), ldx, lb, P(XT), P(XTn), P(ybuf), st))
    chk(lib.ital_chol_append(P(XT), P(XTn), ldx, P(Lc), cap, P(alpha), P(ybuf), 0, 1, var, ls, noise, P(status), st))
    chk(lib.ital_whiten_append(P(Xd), P(xn), n, ldx, P(XT), P(XTn), 1, P(Lc), cap, P(Lc), P(alpha), P(V), ldv, 0, var, ls,
                               P(mu), P(s2), st))
    m = 1
    np.testing.assert_allclose(mu.cpu().numpy(), z["r0_rel_mean"], rtol=0, atol=1e-9)

    # ---- candidate list (reference retrieval_base.py:78-87: ascending, without the labelled sample) and batch state
    cand_h = np.array([i for i in range(n) if i != q], dtype=np.int32)
    nc = len(cand_h)
    cand = torch.from_numpy(cand_h).to(dev)
    alive = torch.ones(nc, dtype=torch.uint8, device=dev)
    mi = torch.zeros(nc, dtype=f64, device=dev)
    bidx, bgpos = torch.zeros(kmax, dtype=i64, device=dev), torch.zeros(kmax, dtype=i64, device=dev)
    bsort = torch.zeros(kmax, dtype=i32, device=dev)
    bmu, sig = torch.zeros(kmax, dtype=f64, device=dev), torch.zeros(kmax * kmax, dtype=f64, device=dev)
    XB, XBn = torch.zeros((kmax, ldx), dtype=f64, device=dev), torch.zeros(kmax, dtype=f64, device=dev)
    VB = torch.zeros((kmax, cap), dtype=f64, device=dev)
    C = torch.zeros((kmax, ldv), dtype=f64, device=dev)
    ret = torch.zeros(kmax + 1, dtype=i64, device=dev)
    rec = torch.zeros(int(lib.ital_record_len(ldx, cap, kmax)), dtype=f64, device=dev)     # sizes from the library, not from prose
    batch = _lib.ItalBatch(kmax, ldx, cap, bidx.data_ptr(), bgpos.data_ptr(), bsort.data_ptr(), bmu.data_ptr(), sig.data_ptr(),
                           XB.data_ptr(), XBn.data_ptr(), VB.data_ptr())

    # ---- the stream of SciPy's mvndst: state and tables from the library
    state = (ctypes.c_int * 6)()
    chk(lib.ital_mvn_seed(state))
    keep = []
    n_alive = nc
    for t in range(1, k + 1):
        desc = _lib.ItalScoreDesc()
        desc.t, desc.n_cand = t, nc
        desc.cand, desc.alive, desc.mu, desc.s2 = cand.data_ptr(), alive.data_ptr(), mu.data_ptr(), s2.data_ptr()
        desc.C, desc.ldc, desc.row_offset, desc.pos_offset, desc.gpos = C.data_ptr(), ldv, 0, 0, None
        desc.batch, desc.noise, desc.eps, desc.label_mode = batch, noise, 1e-12, 0
        desc.mi, desc.status = mi.data_ptr(), status.data_ptr()
        if t >= 3:
            jump = np.empty((_lib.ITAL_JUMP_BITS, 18), dtype=np.int64)
            pat = np.empty((1 << t, 18), dtype=np.int64)
            vk = np.empty(t - 1, dtype=np.float64)
            chk(lib.ital_mvn_tables(t, jump.ctypes.data, pat.ctypes.data, vk.ctypes.data))
            tabs = [torch.from_numpy(a).to(dev) for a in (jump, pat, vk)]
            work = torch.empty(int(lib.ital_round_workspace(t, nc, 0)), dtype=f64, device=dev)
            keep += tabs + [work]
            desc.jump, desc.jumppat, desc.vk = tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr()
            desc.work, desc.work_doubles = work.data_ptr(), work.numel()
            for j in range(6):
                desc.seed[j] = state[j]
        chk(lib.ital_score_step(ctypes.byref(desc), st))
        got_mi = mi.cpu().numpy().copy()
        chk(lib.ital_select_fused(P(mi), P(cand), P(alive), nc, 0, None, 0, 0, 0, P(mu), P(s2), P(Xd), P(xn), ldx, P(V), ldv, m,
                                  cap, P(C), ldv, t - 1, t - 1, batch, P(status), P(rec), P(ret), st))
        if t < k:
            slot = t - 1
            chk(lib.ital_cross_cov_cols(P(Xd), P(xn), n, ldx, P(XB[slot]), P(XBn[slot:]), 1, P(VB[slot]), cap, P(V), ldv, m, var,
                                        ls, P(C[slot]), ldv, st))
        # the reference's serial loop has now made 2 * 2^t mvndst calls per live candidate (ital.py:191-206)
        chk(lib.ital_mvn_advance(state, n_alive * (2 << t) * lib.ital_mvn_draws_per_call(t)))
        # MI vector of this greedy step against the reference's (live positions only)
        want_c, want_mi = z["r0_s%d_cand" % (t - 1)], z["r0_s%d_mi" % (t - 1)]
        pos = {int(c): i for i, c in enumerate(cand_h)}
        mine = got_mi[[pos[int(c)] for c in want_c]]
        np.testing.assert_allclose(mine, want_mi, rtol=1e-8, atol=0)
        n_alive -= 1
    host = ret.cpu().tolist()
    assert host[kmax] == 0 and int(status.item()) == 0
    assert host[:k] == z["r0_ret"].tolist()                    # the reference's batch
    draws = sum((nc - (t - 1)) * (2 << t) * (0 if t < 3 else 8 * (2 * (t - 1) - 1)) for t in range(1, k + 1))
    want_state = (ctypes.c_int * 6)()
    lib.ital_mvn_seed(want_state)
    lib.ital_mvn_advance(want_state, draws)
    assert list(state) == list(want_state)
