"""ctypes binding of libital_hip.so (the C ABI declared in include/ital_hip.h).

The product path has no CPU fallback: if the HIP library is missing this module raises at import.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ITAL_HIP_LIB", os.path.join(HERE, "libital_hip.so"))  # override: kernel-variant experiments

c_void_p, c_int, c_int64, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double

ITAL_MAX_T = 8
ITAL_REC_HEADER = 10
ITAL_JUMP_BITS = 48
ITAL_GENERIC_MAX_DIM = 20
ITAL_GENERIC_MAX_REL = 16
ITAL_GENERIC_MAX_CALLS = 1 << 22
ITAL_TOPK_MAX = 4096
ITAL_ROUND_MAX_CAND = 1 << 21


class ItalBatch(ctypes.Structure):
    _fields_ = [("kmax", c_int), ("ldx", c_int), ("ldw", c_int), ("bidx", c_void_p), ("bgpos", c_void_p),
                ("bsort", c_void_p), ("bmu", c_void_p), ("sig", c_void_p), ("XB", c_void_p), ("XBn", c_void_p),
                ("VB", c_void_p)]


class ItalLabelBatch(ctypes.Structure):
    _fields_ = [("c", c_int), ("slot", c_int * 16), ("y", c_double * 16)]


class ItalAppendDesc(ctypes.Structure):
    _fields_ = [("rows", c_void_p), ("ldx", c_int), ("lb", ItalLabelBatch), ("X", c_void_p), ("xnorm", c_void_p), ("n", c_int64),
                ("XT", c_void_p), ("XTn", c_void_p), ("L", c_void_p), ("ldl", c_int), ("alpha", c_void_p), ("ybuf", c_void_p),
                ("V", c_void_p), ("ldv", c_int64), ("m", c_int), ("var", c_double), ("length_scale", c_double),
                ("noise", c_double), ("mu", c_void_p), ("s2", c_void_p), ("status", c_void_p)]


class ItalScoreDesc(ctypes.Structure):
    _fields_ = [("t", c_int), ("n_cand", c_int64), ("cand", c_void_p), ("alive", c_void_p), ("mu", c_void_p),
                ("s2", c_void_p), ("C", c_void_p), ("ldc", c_int64), ("row_offset", c_int64),
                ("pos_offset", c_int64), ("gpos", c_void_p), ("batch", ItalBatch), ("noise", c_double), ("eps", c_double),
                ("label_mode", c_int), ("mi", c_void_p), ("seed", c_int * 6), ("jump", c_void_p), ("jumppat", c_void_p),
                ("vk", c_void_p), ("status", c_void_p), ("work", c_void_p), ("work_doubles", c_int64), ("ev_start", c_void_p),
                ("ev_stop", c_void_p), ("sel_X", c_void_p), ("sel_xnorm", c_void_p), ("sel_ldx", c_int), ("sel_V", c_void_p),
                ("sel_ldv", c_int64), ("sel_m", c_int), ("sel_ldw", c_int), ("sel_rank", c_int), ("sel_record", c_void_p),
                ("sel_ret", c_void_p), ("sel_parts", c_void_p), ("sel_parts_len", c_int64), ("sel_counter", c_void_p)]


#: int exchange(void* ctx, const double* record, double* records_all, int rec_len, hipStream_t stream): a host's own transport
#: for the record exchange of ital_fetch_round (device pointers as integers)
EXCHANGE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p)


class ItalRoundDesc(ctypes.Structure):
    _fields_ = [("k", c_int), ("step", ItalScoreDesc), ("seeds", (c_int * 6) * (ITAL_MAX_T + 1)),
                ("jump", c_void_p * (ITAL_MAX_T + 1)), ("jumppat", c_void_p * (ITAL_MAX_T + 1)),
                ("vk", c_void_p * (ITAL_MAX_T + 1)), ("ev_start", c_void_p * (ITAL_MAX_T + 1)),
                ("ev_stop", c_void_p * (ITAL_MAX_T + 1)), ("n_rows", c_int64), ("var", c_double), ("length_scale", c_double),
                ("mi_keep", c_void_p), ("begin", c_int), ("cand_prev", c_void_p), ("n_prev", c_int64),
                ("world", c_int), ("records_all", c_void_p), ("nccl_comm", c_void_p), ("exchange", EXCHANGE_FN),
                ("exchange_ctx", c_void_p)]


class ItalGscoreDesc(ctypes.Structure):
    _fields_ = [("n_cand", c_int64), ("cand", c_void_p), ("alive", c_void_p), ("mu", c_void_p), ("s2", c_void_p),
                ("C", c_void_p), ("ldc", c_int64), ("row_offset", c_int64), ("pos_offset", c_int64), ("gpos", c_void_p),
                ("nE", c_int), ("E_idx", c_void_p), ("E_sort", c_void_p), ("E_mu", c_void_p), ("E_sig", c_void_p),
                ("ldE", c_int), ("n_picks", c_int), ("pick_pos", c_void_p), ("subset_mode", c_int), ("fb_mode", c_int),
                ("label_prob", c_double), ("mistake_prob", c_double), ("label_mode", c_int), ("noise", c_double),
                ("eps", c_double), ("clip_cov", c_double), ("seed", c_int * 6), ("jump1", c_void_p), ("vk", c_void_p),
                ("draws_out", c_int64), ("draws_in", c_int64), ("n_in", c_int), ("in_pos", c_void_p),
                ("n_dead", c_int), ("dead_pos", c_void_p), ("mc_rel", c_int), ("rel_samples", c_void_p),
                ("mc_fb", c_int), ("fb_samples", c_void_p), ("draw_off", c_void_p), ("draw_count", c_void_p), ("mi", c_void_p),
                ("status", c_void_p), ("work", c_void_p), ("work_doubles", c_int64), ("pair_count", c_void_p),
                ("defer_join", c_int)]


class ItalMcmiDesc(ctypes.Structure):
    _fields_ = [("t", c_int), ("n_i", c_int64), ("pos_offset", c_int64), ("n_all", c_int64), ("alive", c_void_p),
                ("mu", c_void_p), ("s2", c_void_p), ("cov", c_void_p), ("ld_cov", c_int64), ("C", c_void_p),
                ("ldc", c_int64), ("batch", ItalBatch), ("noise", c_double), ("eps", c_double), ("ce", c_void_p),
                ("work", c_void_p), ("work_doubles", c_int64)]


class ItalMcmiRoundDesc(ctypes.Structure):
    _fields_ = [("k", c_int), ("step", ItalMcmiDesc), ("Xc", c_void_p), ("xnc", c_void_p), ("ldx", c_int), ("Vc", c_void_p),
                ("ldv", c_int64), ("m", c_int), ("ldw", c_int), ("var", c_double), ("length_scale", c_double),
                ("pos", c_void_p), ("status", c_void_p), ("record", c_void_p), ("ret", c_void_p), ("begin", c_int)]


class ItalNpLegacyState(ctypes.Structure):
    _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int32), ("has_gauss", ctypes.c_int32), ("gauss", c_double)]


SIGNATURES = {
    "ital_version": (ctypes.c_char_p, []),
    "ital_mvn_seed": (c_int, [ctypes.POINTER(c_int)]),
    "ital_mvn_draws_per_call": (c_int, [c_int]),
    "ital_mvn_advance": (c_int, [ctypes.POINTER(c_int), c_int64]),
    "ital_mvn_round_seeds": (c_int, [ctypes.POINTER(c_int), c_int64, c_int, c_void_p]),
    "ital_mvn_tables": (c_int, [c_int, c_void_p, c_void_p, c_void_p]),
    "ital_mvn_generic_tables": (c_int, [c_int, c_void_p, c_void_p]),
    "ital_np_legacy_normals": (c_int, [ctypes.POINTER(ItalNpLegacyState), c_int64, c_void_p, c_int64, c_int]),
    "ital_last_error": (ctypes.c_char_p, []),
    "ital_launch_count": (c_int64, []),
    "ital_row_norms": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "ital_rbf_cols": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_double, c_double,
                              c_void_p, c_int64, c_void_p]),
    "ital_cross_cov_cols": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                    c_void_p, c_int64, c_int, c_double, c_double, c_void_p, c_int64, c_void_p]),
    "ital_chol_append": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                 c_double, c_double, c_double, c_void_p, c_void_p]),
    "ital_whiten_append": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p, c_int64, c_int, c_double, c_double, c_void_p,
                                   c_void_p, c_void_p]),
    "ital_stage_labelled": (c_int, [c_void_p, c_int, ItalLabelBatch, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ital_gp_append": (c_int, [ctypes.POINTER(ItalAppendDesc), c_void_p]),
    "ital_predict": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                             c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "ital_topk": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ital_topk_workspace": (c_int64, []),
    "ital_score_step": (c_int, [ctypes.POINTER(ItalScoreDesc), c_void_p]),
    "ital_score_workspace": (c_int64, [c_int, c_int64]),
    "ital_fetch_round": (c_int, [ctypes.POINTER(ItalRoundDesc), c_void_p]),
    "ital_cov_block": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64,
                               c_void_p, c_int64, c_int, c_double, c_double, c_void_p, c_int64, c_void_p]),
    "ital_cov_abs_rowsum": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64,
                                    c_void_p, c_int64, c_int, c_double, c_double, c_void_p, c_int64, c_int, c_void_p,
                                    c_void_p]),
    "ital_mcmi_score_step": (c_int, [ctypes.POINTER(ItalMcmiDesc), c_void_p]),
    "ital_mcmi_workspace": (c_int64, [c_int, c_int64]),
    "ital_score_generic": (c_int, [ctypes.POINTER(ItalGscoreDesc), c_void_p]),
    "ital_score_generic_workspace": (c_int64, [ctypes.POINTER(ItalGscoreDesc)]),
    "ital_select_local": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p,
                                  c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ital_select_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p,
                                  c_int64, c_int, c_int, ItalBatch, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ital_select_exchange": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ital_exchange_info": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_char_p, c_int]),
    "ital_exchange_error": (c_int, [c_void_p, c_void_p]),
    "ital_score_generic_join": (c_int, [c_void_p]),
    "ital_ctx_create": (c_int, [c_int64, c_int, c_double, c_double, c_double, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ital_ctx_destroy": (c_int, [c_void_p]),
    "ital_ctx_fit": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "ital_ctx_update": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ital_ctx_fetch": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "ital_ctx_predict_stored": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "ital_ctx_local_rows": (c_int64, [c_void_p, c_void_p]),
    "ital_record_len": (c_int, [c_int, c_int, c_int]),
    "ital_round_workspace": (c_int64, [c_int, c_int64, c_int64]),
    "ital_sel_parts_len": (c_int64, [c_int, c_int64, c_int64]),
    "ital_mcmi_round": (c_int, [ctypes.POINTER(ItalMcmiRoundDesc), c_void_p]),
    "ital_gather_block": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ital_select_resolve": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, ItalBatch, c_void_p, c_void_p,
                                    c_void_p]),
}


class ItalHipError(RuntimeError):
    pass


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -m ital_amd.build` (hipcc, gfx950). "
            "ital_amd has no CPU fallback.")
    # The library links against libamdhip64 by soname.  A PyTorch host brings its OWN copy of the HIP runtime; whichever is
    # loaded first serves both -- and two runtimes in one process means kernels registered with one and streams created by
    # the other ("no ROCm-capable device is detected" at the first launch: seen in round 5 when build() had loaded this
    # library before smoke() imported torch).  So torch goes first whenever it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(rc):
    if rc != 0:
        raise ItalHipError(f"libital_hip: error {rc}: {lib().ital_last_error().decode()}")


def legacy_normals(n_skip, n_out, threads=1):
    """Advances numpy's global legacy generator past `n_skip` standard normals without computing them, then draws the next
    `n_out` (float64 array) -- the same values, and the same generator state afterwards, as
    `np.random.standard_normal(n_skip); np.random.standard_normal(n_out)` (ital_np_legacy_normals; `threads` host threads
    produce the values of a large request)."""
    import numpy as np
    n_skip, n_out = int(n_skip), int(n_out)
    state = np.random.get_state(legacy=True)
    if state[0] != "MT19937":                 # somebody replaced the global bit generator: numpy walks its own stream
        if n_skip:
            np.random.standard_normal(n_skip)
        return np.random.standard_normal(n_out)
    st = ItalNpLegacyState()
    key = np.ascontiguousarray(state[1], dtype=np.uint32)
    ctypes.memmove(st.key, key.ctypes.data, 624 * 4)
    st.pos, st.has_gauss, st.gauss = int(state[2]), int(state[3]), float(state[4])
    out = np.empty(n_out, dtype=np.float64)
    check(lib().ital_np_legacy_normals(ctypes.byref(st), n_skip, out.ctypes.data, n_out, int(threads)))
    np.random.set_state(("MT19937", np.frombuffer(st.key, dtype=np.uint32).copy(), int(st.pos), int(st.has_gauss),
                         float(st.gauss)))
    return out
