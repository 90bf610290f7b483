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


# Keast's Korobov generators C(NP, NDIM-1) of Genz's MVNDST for NDIM = 2..19 (the tests' own copy: the library's table is
# checked against it, and against SciPy through the oracle's restatement in test_oracle_mvndst.py)
KOROBOV_C = {2: 13, 3: 28, 4: 27, 5: 28, 6: 20, 7: 92, 8: 102, 9: 339, 10: 206, 11: 422, 12: 134, 13: 518, 14: 134,
             15: 134, 16: 518, 17: 652, 18: 382, 19: 206}


def test_korobov_generators():
    for n in range(3, 21):
        vk = ms.korobov_vk(n)
        p = ms.PRIMES[min(n - 1, 10) - 1]
        want = np.empty(n - 1)
        want[0] = 1.0 / p
        for i in range(1, n - 1):
            want[i] = np.fmod(float(KOROBOV_C[n - 1]) * want[i - 1], 1.0)
        assert np.array_equal(vk, want)
        assert abs(vk[1] - (KOROBOV_C[n - 1] % p) / p) < 1e-15


def test_stream_entry_points_of_the_c_abi():
    """ital_mvn_seed / _advance / _draws_per_call / _tables / _generic_tables driven through ctypes alone (what a host that
    is not Python binds, include/ital_hip.h) against the oracle's generator."""
    from ital_amd import _lib
    lib = _lib.load()
    st = (ctypes.c_int * 6)()
    assert lib.ital_mvn_seed(st) == 0 and tuple(st) == ms.SEED
    mvn.rng_reset()
    total = 0
    for n in (1, 7, 40, 123456789, 3):
        mvn.rng_skip(n)
        assert lib.ital_mvn_advance(st, n) == 0
        total += n
        assert list(st) == mvn.rng_state()
    assert lib.ital_mvn_advance(st, 0) == 0 and list(st) == mvn.rng_state()
    assert lib.ital_mvn_advance(st, -1) != 0 and b"ital_mvn_advance" in lib.ital_last_error()
    assert [lib.ital_mvn_draws_per_call(n) for n in (1, 2, 3, 4, 9)] == [0, 0, 24, 40, 120]
    jump = np.zeros((48, 18), dtype=np.int64)
    pat = np.zeros((16, 18), dtype=np.int64)
    vk = np.zeros(3)
    assert lib.ital_mvn_tables(4, jump.ctypes.data, pat.ctypes.data, vk.ctypes.data) == 0
    calls = 0b110101
    cur = list(ms.SEED)
    for b in range(6):
        if (calls >> b) & 1:
            cur = _apply(jump[b], cur)
    mvn.rng_reset()
    mvn.rng_skip(calls * 40)
    assert cur == mvn.rng_state()
    mvn.rng_reset()
    mvn.rng_skip(2 * 11 * 40)
    assert _apply(pat[11], list(ms.SEED)) == mvn.rng_state() and _apply(pat[0], list(ms.SEED)) == list(ms.SEED)
    assert np.array_equal(vk, ms.korobov_vk(4))
    assert lib.ital_mvn_tables(2, None, None, None) != 0 and lib.ital_mvn_tables(9, None, pat.ctypes.data, None) != 0
    assert lib.ital_mvn_generic_tables(21, None, None) != 0


def test_legacy_normals_skip_and_fill_are_numpy_bit_for_bit():
    """ital_np_legacy_normals against np.random.standard_normal: values and generator state, for every parity of
    (cached value, skip count, fill count), across MT19937 block boundaries, single- and multi-threaded."""
    from ital_amd import _lib
    for seed, pre in ((0, 0), (1, 3), (2, 624 * 2 + 1)):
        for skip, fill, th in ((0, 7, 1), (1, 10, 1), (2, 1001, 1), (5, 0, 1), (12345, 54321, 1), (1872, 5, 1),
                               (77, 200001, 4), (1000000, 3, 1)):
            np.random.seed(seed)
            np.random.standard_normal(pre)            # an odd count leaves a cached second value behind
            start = np.random.get_state()
            np.random.standard_normal(skip)
            want = np.random.standard_normal(fill)
            tail = np.random.standard_normal(5)
            np.random.set_state(start)
            got = _lib.legacy_normals(skip, fill, th)
            assert np.array_equal(got, want), (seed, pre, skip, fill)
            assert np.array_equal(np.random.standard_normal(5), tail), (seed, pre, skip, fill)
    np.random.seed(5)
    a = np.random.uniform(size=3)                     # the uniform stream is the same generator: still aligned
    np.random.seed(5)
    _lib.legacy_normals(0, 0)
    assert np.array_equal(np.random.uniform(size=3), a)


def test_c_abi_exports_every_declared_symbol():
    from ital_amd import _lib
    header = open(os.path.join(ROOT, "include", "ital_hip.h")).read()
    declared = set(re.findall(r"\b(ital_[a-z_0-9]+)\s*\(", header))
    declared -= {"ital_batch", "ital_score_desc", "ital_round_desc", "ital_append_desc", "ital_mcmi_round_desc"}
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


def _bare_ital(**kw):
    """An ITAL instance without a device (only the option logic is exercised)."""
    from ital_amd.ital import ITAL
    L = object.__new__(ITAL)
    opts = dict(label_prob=1.0, mistake_prob=0.0, top_candidates=None, change_estimation_subset=0, clip_cov=0,
                label_estimation="mean", monte_carlo_num_rel=None, monte_carlo_num_fb=None, force_generic=False)
    opts.update(kw)
    for k, v in opts.items():
        setattr(L, k, v)
    return L


def test_monte_carlo_plan_follows_the_reference_thresholds():
    """rel_iter samples once 2^(n-1) >= n*mc (ital.py:293-295); fb_iter once 2^(n-1) >= n*mc for the motivated user
    (:318-320) and 3^n >= 2*n*mc for the general one (:331-333)."""
    L = _bare_ital(monte_carlo_num_rel=2, monte_carlo_num_fb=2)
    assert [L._mc_plan(n, 0)[:2] for n in (1, 3, 4, 5)] == [(False, 2), (False, 8), (True, 8), (True, 10)]
    assert [L._mc_plan(n, 1)[2:] for n in (1, 3, 4)] == [(False, 2), (False, 8), (True, 8)]
    assert [L._mc_plan(n, 2)[2:] for n in (1, 2, 3)] == [(False, 2), (True, 4), (True, 6)]
    assert _bare_ital()._mc_plan(5, 2) == (False, 32, False, 242)


def test_unsupported_option_combinations_are_named():
    assert _bare_ital()._unsupported(8) is None
    assert "monte_carlo_num_rel" in _bare_ital()._unsupported(9)
    assert _bare_ital(monte_carlo_num_rel=1)._unsupported(16) is None
    assert "larger than 16" in _bare_ital(monte_carlo_num_rel=1)._unsupported(17)
    assert _bare_ital(change_estimation_subset=5)._unsupported(4) is None
    assert "dimension" in _bare_ital(change_estimation_subset=18)._unsupported(4)
    assert "change_estimation_subset=None" in _bare_ital(change_estimation_subset=None)._unsupported(2, 500)
    assert _bare_ital(change_estimation_subset=None)._unsupported(2, 15) is None
    assert _bare_ital(clip_cov=0.5)._unsupported(6) is None and _bare_ital(clip_cov=0.5)._needs_generic()
    assert not _bare_ital(clip_cov=1.5)._needs_generic()                    # outside (0, 1): no effect (ital.py:360)
    assert "label_estimation" in _bare_ital(label_estimation="median")._unsupported(2)
    assert "orthant probabilities per candidate" in _bare_ital(label_prob=0.5)._unsupported(12)
    assert _bare_ital(label_prob=0.5, monte_carlo_num_rel=1, monte_carlo_num_fb=1)._unsupported(12) is None


def test_stream_tables_are_consistent():
    """jump tables: 2^b calls, the prior call of every sign pattern and single draws describe the same generator."""
    from ital_amd import mvn_stream as ms
    n = 5
    d = ms.draws_per_call(n)
    s = ms.MvnStream()
    base = s.state
    jl = ms.jump_pattern_table(n)
    j2 = ms.jump_table(n, 8)
    j1 = ms.jump1_table(16)

    def apply(row, st):
        a = [sum(int(row[3 * i + k]) * st[k] for k in range(3)) % ms.M1 for i in range(3)]
        b = [sum(int(row[9 + 3 * i + k]) * st[3 + k] for k in range(3)) % ms.M2 for i in range(3)]
        return tuple(a + b)

    assert apply(jl[0], base) == base and len(jl) == 2 ** n
    assert apply(jl[5], base) == s.peek(10 * d) and apply(jl[31], base) == s.peek(62 * d)
    assert apply(j2[3], base) == s.peek(8 * d)
    st = base
    for bit in range(16):
        if (37 >> bit) & 1:
            st = apply(j1[bit], st)
    assert st == s.peek(37)
    vk = ms.vk_table(6)
    assert np.all(vk[:3] == 0) and np.array_equal(vk[5, :4], ms.korobov_vk(5))


def test_ctypes_structs_have_the_layout_of_the_header(tmp_path):
    """sizeof / offsetof of every descriptor as gcc lays out include/ital_hip.h against the ctypes mirrors."""
    import ctypes
    import subprocess
    from ital_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probes = {"ital_batch": (_lib.ItalBatch, ["kmax", "bidx", "VB"]),
              "ital_label_batch": (_lib.ItalLabelBatch, ["c", "slot", "y"]),
              "ital_score_desc": (_lib.ItalScoreDesc, ["t", "gpos", "batch", "label_mode", "seed", "jumppat", "status", "work", "work_doubles", "ev_stop", "sel_ldx", "sel_ldv", "sel_rank", "sel_ret", "sel_parts_len",
                                                       "sel_counter"]),
              "ital_gscore_desc": (_lib.ItalGscoreDesc, ["n_cand", "gpos", "nE", "ldE", "pick_pos", "label_prob", "clip_cov", "seed",
                                                         "draws_in", "dead_pos", "fb_samples", "draw_count", "status", "work_doubles", "pair_count"]),
              "ital_mcmi_desc": (_lib.ItalMcmiDesc, ["t", "alive", "ld_cov", "batch", "ce", "work", "work_doubles"]),
              "ital_round_desc": (_lib.ItalRoundDesc, ["k", "step", "seeds", "jump", "vk", "ev_stop", "n_rows", "length_scale",
                                                       "mi_keep", "begin", "cand_prev", "n_prev", "world", "records_all", "nccl_comm", "exchange",
                                                       "exchange_ctx"]),
              "ital_mcmi_round_desc": (_lib.ItalMcmiRoundDesc, ["k", "step", "Xc", "ldx", "Vc", "ldv", "m", "ldw", "var", "pos",
                                                                "status", "record", "ret", "begin"]),
              "ital_np_legacy_state": (_lib.ItalNpLegacyState, ["key", "pos", "has_gauss", "gauss"]),
              "ital_append_desc": (_lib.ItalAppendDesc, ["rows", "lb", "X", "n", "ldl", "ybuf", "ldv", "m", "noise", "status"])}
    lines = []
    for name, (_, fields) in probes.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for f in fields:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ital_hip.h"\nint main(void) {\n' + "\n".join(lines)
                   + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, (cls, fields) in probes.items():
        assert int(got[name]) == ctypes.sizeof(cls), name
        for f in fields:
            assert int(got["%s.%s" % (name, f)]) == getattr(cls, f).offset, (name, f)


def test_score_workspace_size():
    """ital_score_workspace (host-only entry point): records of 2^t patterns, their terms and the generator state."""
    from ital_amd import _lib
    lib = _lib.load()
    for t, lat in ((3, 10), (4, 15), (8, 35)):      # lattices packed: 5 (t - 1) doubles per call
        ncor = t * (t - 1) // 2
        per_cand = (1 << t) * (ncor + t + 1 + lat + 1) + 3
        assert lib.ital_score_workspace(t, 1) == per_cand
        assert lib.ital_score_workspace(t, 1000) == 1000 * per_cand
    assert lib.ital_score_workspace(2, 10) == 0 and lib.ital_score_workspace(9, 10) == 0
    assert lib.ital_topk_workspace() > 4096 * 8


def test_host_side_under_address_and_ub_sanitizers():
    """tools/asan_host.sh: the library's host code (argument validation of every entry point, descriptor handling, the
    stream bookkeeping, the walker of numpy's generator, the RCCL lookup) built with -fsanitize=address,undefined and driven
    by tests/host_asan_driver.cpp on the CPU.  (GPU ASan is not available on the pool; the device code is never launched.)"""
    import subprocess
    res = subprocess.run([os.path.join(ROOT, "tools", "asan_host.sh")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "OK host side under ASan + UBSan" in res.stdout


def test_rank_local_walk_of_numpy_stream_partitions_the_reference_draws():
    """What the Monte-Carlo pattern sampler does on W ranks (ital_amd/ital.py `_walk_normals`): every rank walks numpy's
    global generator over ALL candidates' standard normals, computing only those of its own slice.  The slices put together
    are the reference's draws (one multivariate_normal.rvs per candidate in list order, ital.py:297), and every rank's
    generator ends in the reference's state."""
    from ital_amd import _lib, sharding
    n_live, per_cand, world = 1001, 3 * 3, 4
    np.random.seed(42)
    np.random.standard_normal(5)                      # some earlier consumption, cached value pending
    start = np.random.get_state()
    want = np.random.standard_normal(n_live * per_cand)
    end_tail = np.random.standard_normal(4)
    got = []
    for rank in range(world):
        j0, j1 = sharding.row_range(n_live, world, rank)
        np.random.set_state(start)
        z = _lib.legacy_normals(j0 * per_cand, (j1 - j0) * per_cand, threads=2)
        _lib.legacy_normals((n_live - j1) * per_cand, 0)
        assert np.array_equal(np.random.standard_normal(4), end_tail)
        got.append(z)
    assert np.array_equal(np.concatenate(got), want)


def test_contiguous_runs_is_a_function_of_the_list_alone():
    from ital_amd import sharding
    n, world = 1000, 4
    asc = np.delete(np.arange(n), [3, 500, 999])
    assert sharding.contiguous_runs(asc, n, world)
    for r in range(world):                            # ... and then no rank gets explicit positions
        r0, r1 = sharding.row_range(n, world, r)
        assert sharding.shard_candidates(asc, r0, r1)[2] is None
    shuffled = np.random.default_rng(0).permutation(asc)[:40]      # the argpartition order of top_candidates
    assert not sharding.contiguous_runs(shuffled, n, world)
    assert sharding.contiguous_runs(shuffled, n, 1)
    assert sharding.contiguous_runs(np.array([10, 11, 600, 990]), n, world)


def test_round_entry_points_refuse_bad_descriptors_without_touching_the_device():
    """Argument checks of the one-call entry points happen before any HIP call: usable (and tested) without a GPU.  Every
    refusal is -22 with a message that names the entry point."""
    import ctypes
    from ital_amd import _lib
    lib = _lib.lib()

    def refused(rc, name):
        assert rc == -22
        assert name in lib.ital_last_error().decode()

    refused(lib.ital_fetch_round(None, None), "ital_fetch_round")
    r = _lib.ItalRoundDesc()
    r.k = 0
    refused(lib.ital_fetch_round(ctypes.byref(r), None), "ital_fetch_round")           # k outside 1 .. 8
    r.k = 4
    r.step.batch.kmax = 4
    r.step.n_cand = 2
    refused(lib.ital_fetch_round(ctypes.byref(r), None), "ital_fetch_round")           # fewer candidates than steps
    r.step.n_cand = 100
    refused(lib.ital_fetch_round(ctypes.byref(r), None), "ital_fetch_round")           # sel_* missing
    r.world = 2                                                                           # several ranks without a transport
    refused(lib.ital_fetch_round(ctypes.byref(r), None), "ital_fetch_round")
    r.world = 0
    r.step.n_cand = (1 << 18) + 1
    refused(lib.ital_fetch_round(ctypes.byref(r), None), "ital_fetch_round")           # beyond the single-call limit

    refused(lib.ital_mcmi_round(None, None), "ital_mcmi_round")
    m = _lib.ItalMcmiRoundDesc()
    m.k = 9
    refused(lib.ital_mcmi_round(ctypes.byref(m), None), "ital_mcmi_round")
    m.k = 2
    m.step.batch.kmax = 2
    m.step.n_i, m.step.n_all, m.step.pos_offset = 10, 12, 0
    refused(lib.ital_mcmi_round(ctypes.byref(m), None), "ital_mcmi_round")             # one rank scores the whole block
    m.step.n_all = 10
    refused(lib.ital_mcmi_round(ctypes.byref(m), None), "ital_mcmi_round")             # null buffers

    refused(lib.ital_gather_block(None, 5, 0, 10, None, None, 16, None, 16, 0, None, None, None, None, 16, None, None, None, None),
            "ital_gather_block")
    assert lib.ital_gather_block(None, 0, 0, 10, None, None, 16, None, 16, 0, None, None, None, None, 16, None, None, None, None) == 0
    refused(lib.ital_select_exchange(None, None, 0, None, None), "ital_select_exchange")


def test_all_upper_form_of_the_lattice_sums_is_an_identity_and_keeps_the_references_roundings():
    """The arithmetic behind ITAL_QMC_FLIP (csrc/qmc_common.h), checked on the host: a variable bounded below enters the
    lattice sums negated.  (1) Phi^-1(d + x (1 - d)), d = Phi(z), equals -Phi^-1((1 - x) Phi(-z)); (2) the lattice
    coordinate of the negated variable, |2 frac(v + 1/2) - 1|, is 1 - |2 frac(v) - 1|; (3) the interval width the reference
    forms, 1 - Phi(z) with the ROUNDED Phi(z) = 1 - p, is what flip_width rebuilds from p = Phi(-z): 1 - (1 - p) -- exactly 0
    iff p <= 2^-54, a multiple of 2^-53 below 1/2, and the identity for widths >= 1/2."""
    from scipy.special import ndtr, ndtri
    rng = np.random.default_rng(12)
    z = rng.uniform(-3, 3, 2000)        # (further out the reference's own form d + x (1 - d) loses digits next to 1:
    x = rng.uniform(0.001, 0.999, 2000)  #  1e-7 in Phi^-1 at z = 6 -- the negated form does not)
    d = ndtr(z)
    np.testing.assert_allclose(ndtri(d + x * (1 - d)), -ndtri((1 - x) * ndtr(-z)), rtol=1e-9, atol=1e-9)
    v = rng.uniform(0, 1400, 5000)
    fr = lambda t: t - np.floor(t)
    np.testing.assert_allclose(np.abs(2 * fr(v + 0.5) - 1), 1 - np.abs(2 * fr(v) - 1), rtol=0, atol=1e-12)
    p = np.array([0.0, 2.0 ** -60, 2.0 ** -54, np.nextafter(2.0 ** -54, 1), 2.0 ** -53, 1e-16, 3e-16, 1e-9, 0.3])
    w = 1.0 - (1.0 - p)
    assert list(w[:3]) == [0.0, 0.0, 0.0] and np.all(w[3:] > 0)
    assert np.all(w / 2.0 ** -53 == np.round(w / 2.0 ** -53))            # on the grid of doubles below 1
    big = rng.uniform(0.5, 1.0, 1000)
    assert np.array_equal(1.0 - (1.0 - big), big)


def test_unseen_list_equals_the_rebuilt_array_under_random_removals():
    """retrieval_base.UnseenList (base + removed ids, O(k log N) per round) against the list rebuilt from scratch as the
    reference does on every call (retrieval_base.py:78-87): length, positions of row boundaries, the entries of a row
    range, the materialised array -- through removals, failed removals and a rebase."""
    from ital_amd.retrieval_base import UnseenList
    rng = np.random.default_rng(3)
    n = 5000
    seen0 = rng.choice(n, 40, replace=False)
    want = np.setdiff1d(np.arange(n), seen0)
    u = UnseenList(want.copy())
    u.REBASE_AT = 64
    version = 0
    for step in range(60):
        ids = rng.choice(want, int(rng.integers(1, 9)), replace=False)
        if step % 7 == 3:
            assert not u.remove(list(ids) + [int(seen0[0])])          # one id is not on the list: nothing changes
            assert not u.remove([n + 5]) and u.version == version
        assert u.remove(ids.tolist())
        version += 1
        want = np.setdiff1d(want, ids)
        assert u.version == version and u.last_removed == tuple(sorted(int(i) for i in ids))
        assert not u.remove([int(ids[0])])                           # already taken out
        assert len(u) == len(want)
        for x in (0, 1, int(want[7]), int(ids[0]), int(ids[0]) + 1, n // 2, n - 1, n, n + 10):
            assert u.count_below(x) == int(np.searchsorted(want, x))
        r0, r1 = sorted(int(v) for v in rng.integers(0, n + 1, 2))
        np.testing.assert_array_equal(u.in_rows(r0, r1), want[(want >= r0) & (want < r1)])
        if step % 3 == 0:
            arr = u.array()
            np.testing.assert_array_equal(arr, want)
            assert u.array() is arr                                  # the same object until the next removal
    assert len(u.removed) < 64 + 9                                   # materialising rebased the list on the way
    np.testing.assert_array_equal(u.array(), want)


def test_size_helpers_of_the_c_abi():
    """ital_record_len / ital_round_workspace / ital_sel_parts_len (host-only entry points, round 5): the buffer sizes a host
    owns, as functions instead of prose in the header (reference: the buffers stand in for the per-process state of
    ital/ital.py:504-529)."""
    from ital_amd import _lib
    lib = _lib.load()
    assert lib.ital_record_len(256, 32, 4) == _lib.ITAL_REC_HEADER + 256 + 32 + 4
    assert lib.ital_record_len(-1, 0, 0) == 0
    per4, per8 = lib.ital_score_workspace(4, 1), lib.ital_score_workspace(8, 1)
    # one slab when it fits, else the cap, never below one candidate
    assert lib.ital_round_workspace(4, 9298, 0) == 9298 * per4
    assert lib.ital_round_workspace(4, 9298, 1 << 27) == 9298 * per4
    assert lib.ital_round_workspace(8, 25000, 1 << 27) == 1 << 27
    assert lib.ital_round_workspace(8, 25000, 10) == per8
    assert lib.ital_round_workspace(2, 1000, 0) == 0
    # three doubles per block of the largest scoring launch (n / 32 at t = 2) + 64 + one per slab
    assert lib.ital_sel_parts_len(2, 9298, 0) == 3 * (9298 // 32 + 64)
    assert lib.ital_sel_parts_len(4, 9298, 9298 * per4) == 3 * (9298 // 32 + 64 + 1)
    slabs = -(-25000 // ((1 << 27) // per8))
    assert lib.ital_sel_parts_len(8, 25000, 1 << 27) == 3 * (25000 // 32 + 64 + slabs)
    assert lib.ital_sel_parts_len(8, 25000, 1) == 3 * (25000 // 32 + 64 + 25000)      # less than one candidate: a slab each


def test_id_sets_count_their_in_place_changes():
    """relevant_ids / irrelevant_ids / unnameable_ids are public sets as in the reference (retrieval_base.py:50-52); the learners
    keep them as IdSet so that ANY in-place change -- also one that leaves the size alone -- invalidates the kept candidate list."""
    import copy
    import pickle
    from ital_amd.retrieval_base import IdSet
    s = IdSet([1, 2])
    c0 = s.changes                                                  # (a process-unique start value per wrapper, see below)
    assert s == {1, 2} and isinstance(s, set)
    s.add(3); s |= {4}; s.discard(1); s.remove(2); s.update([7, 8])
    assert s == {3, 4, 7, 8} and s.changes == c0 + 5
    assert type(s | {9}) is set and s.changes == c0 + 5             # non-mutating operators do not count
    s.pop(); s.clear()
    assert s.changes == c0 + 7 and len(s) == 0
    t = IdSet([5])
    assert copy.deepcopy(t) == {5} and pickle.loads(pickle.dumps(t)) == {5}
    # round-5 advice: re-wrapping (learner.relevant_ids = learner.relevant_ids.copy(): a plain set again) must never land on a
    # token an earlier wrapper held -- every wrapper starts its counter at a value of its own, far from the others'
    u, v = IdSet([5]), IdSet([6])
    assert u.changes != v.changes and abs(u.changes - v.changes) >= 1 << 20 and t.changes not in (u.changes, v.changes)


def test_ranges_of_a_sampled_step_grow_geometrically():
    """ital_amd.ital._range_cuts (round 5): a step of sampled patterns (reference ital.py:293-297) is scored in ranges of its
    candidates so that the host's per-candidate decompositions of range r + 1 run under the GPU's lattice sums of range r;
    the first range -- decomposed while the GPU idles -- is short, the sizes double."""
    from ital_amd.ital import _range_cuts
    c = _range_cuts(0, 1_000_000, 6)
    assert c[0] == 0 and c[-1] == 1_000_000 and len(c) == 7
    sizes = np.diff(c)
    assert sizes[0] <= 1_000_000 // 60 and np.all(sizes[1:] >= 2 * sizes[:-1] - 2)          # 1 : 2 : 4 : ...
    c = _range_cuts(5, 33_000, 4)                                # a short span: ranges below the minimum are merged
    assert c[0] == 5 and c[-1] == 33_000 and np.all(np.diff(c) >= 8192)
    assert _range_cuts(10, 2_010, 4).tolist() == [10, 2_010]    # shorter than the minimum: one range
    assert _range_cuts(0, 125_000, 4).tolist() == [0, 8333, 25000, 58333, 125000]


def test_sampled_cpu_baseline_of_the_bench():
    """bench.py's extrapolated CPU baselines (C3' - C5' "timed on a candidate subsample"): the oracle in the reference's
    fork-pool scheme (ital/ital.py:124-126) with a pilot of one candidate per worker and as many more as the step's share of the
    budget allows.  With a generous budget every live candidate of every step is scored -- the step's pick is then the serial
    oracle's for the closed-form steps (t <= 2) -- and the cores it runs on are those the process may really use."""
    from oracle.ital import OracleITAL
    from oracle.parallel import effective_cores, fetch_unlabelled_sampled
    assert 1 <= effective_cores() <= (os.cpu_count() or 1)
    X = np.random.default_rng(5).random((60, 6))
    A = OracleITAL(X, length_scale=0.7)
    A.update({0: 1, 1: -1})
    steps = fetch_unlabelled_sampled(A, 2, processes=2, budget_s=60.0, n_max=1000)
    assert [s["t"] for s in steps] == [1, 2] and [s["scored"] for s in steps] == [58, 57]
    assert all(s["per_cand_s"] > 0 and s["fork_s"] >= 0 and abs(s["per_cand_s"] * s["scored"] - s["map_s"]) < 1e-9 for s in steps)
    # a tight budget: the pilot (one candidate per worker) is the least a step scores
    B = OracleITAL(X, length_scale=0.7)
    B.update({0: 1, 1: -1})
    steps = fetch_unlabelled_sampled(B, 3, processes=2, budget_s=0.0, n_max=1000)
    assert [s["scored"] for s in steps] == [2, 2, 2]
