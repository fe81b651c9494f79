"""ctypes binding of liba4nt_hip.so (declared in include/a4nt_hip.h).

The product path has no CPU fallback: a missing library or a non-gfx950 device raises.
"""
import ctypes
import os
from ctypes import c_int, c_uint, c_float, c_longlong, c_ulonglong, c_void_p, c_char_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# A4NT_HIP_LIB: diagnostic builds only (tools/lstm_stamp.py loads the phase-stamp variant)
LIB_PATH = os.environ.get('A4NT_HIP_LIB') or os.path.join(_HERE, 'liba4nt_hip.so')

P = c_void_p


class LstmFwdProblem(ctypes.Structure):      # a4nt_lstm_fwd_problem
    _fields_ = [('T', c_int), ('B', c_int)] + [(n, c_void_p) for n in (
        'gx', 'gc', 'whh', 'h0', 'c0', 'lens', 'hs', 'gates', 'cs', 'hN', 'cN', 'xch', 'sync')]


class LstmBwdProblem(ctypes.Structure):      # a4nt_lstm_bwd_problem
    _fields_ = [('T', c_int), ('B', c_int)] + [(n, c_void_p) for n in (
        'whh', 'gates', 'cs', 'c0', 'lens', 'dhs', 'dhN', 'dcN', 'dgates', 'dh0', 'dc0', 'xch', 'sync', 'dbias')]


class GenFwdArgs(ctypes.Structure):          # a4nt_gen_fwd_args
    _fields_ = [('B', c_int), ('H', c_int), ('V', c_int), ('n_max', c_int),
                ('cg', c_void_p), ('ptab', c_void_p), ('tok0', c_void_p), ('ones', c_void_p), ('whh', c_void_p),
                ('hs_all', c_void_p), ('gates', c_void_p), ('cs', c_void_p), ('xch', c_void_p),
                ('W', c_void_p), ('ldw', c_int), ('bias', c_void_p), ('scale', c_float), ('tau', c_float),
                ('hard', c_int), ('greedy', c_int),
                ('U', POINTER(c_void_p)), ('n_noise', c_int), ('ldu', c_int),
                ('seed', c_ulonglong), ('step0', c_uint),
                ('z', c_void_p), ('ldz', c_int), ('zrow0', c_int), ('zrows', c_int),
                ('part', c_void_p), ('stats', c_void_p), ('tok', c_void_p), ('val_in', c_void_p), ('val_out', c_void_p),
                ('gen_lens', c_void_p), ('done', c_void_p), ('end_c', c_longlong)]


class GenBwdArgs(ctypes.Structure):          # a4nt_gen_bwd_args
    _fields_ = [('n', c_int), ('B', c_int), ('H', c_int), ('E', c_int), ('V', c_int),
                ('S', c_void_p), ('lds', c_int), ('D', c_void_p), ('ldd', c_int), ('y', c_void_p), ('ldy', c_int),
                ('stats', c_void_p), ('coef', c_float), ('inv_tau', c_float),
                ('emb_W', c_void_p), ('W_out', c_void_p), ('ldw', c_int), ('Wcat', c_void_p),
                ('gates', c_void_p), ('cs', c_void_p), ('dc', c_void_p), ('dG', c_void_p), ('rec', c_void_p),
                ('dh_slabs', c_void_p), ('ks_v', c_int), ('rec_slabs', c_void_p), ('ks_r', c_int),
                ('fold', c_void_p), ('abuf', c_void_p)]


_SIGS = {
    'a4nt_last_error': (c_char_p, []),
    'a4nt_version': (c_int, []),
    'a4nt_init': (c_int, [c_int, POINTER(c_int)]),
    'a4nt_sgemm': (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int,
                           c_longlong, P]),
    'a4nt_sgemm_tile': (c_int, [c_int, c_int, c_int]),
    'a4nt_sgemm_set_kgroups': (c_int, [c_int]),
    'a4nt_sgemm_nt_rows_ok': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'a4nt_sgemm_nt_rows': (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'a4nt_sgemm_skinny_ksplit': (c_int, [c_int, c_int, c_int]),
    'a4nt_sgemm_skinny': (c_int, [c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_longlong, P]),
    'a4nt_lstm_geometry': (c_int, [c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'a4nt_lstm_set_max_workgroups': (c_int, [c_int]),
    'a4nt_lstm_resident_workgroups': (c_int, [c_int, c_int, c_int]),
    'a4nt_lstm_xch_floats': (c_longlong, [c_int, c_int, c_int]),
    'a4nt_lstm_sync_words': (c_int, [c_int, c_int]),
    'a4nt_lstm_seq_fwd': (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'a4nt_lstm_seq_fwd_noalloc': (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'a4nt_lstm_seq_bwd': (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'a4nt_lstm_bwd_bias_rows': (c_int, [c_int, c_int]),
    'a4nt_lstm_seq_bwd_bias': (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'a4nt_lstm_pair_geometry': (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'a4nt_lstm_pair_xch_floats': (c_longlong, [c_int, c_int, c_int, c_int]),
    'a4nt_lstm_pair_sync_words': (c_int, [c_int, c_int]),
    'a4nt_lstm_seq_fwd_pair': (c_int, [c_int, POINTER(LstmFwdProblem), POINTER(LstmFwdProblem), P]),
    'a4nt_lstm_seq_bwd_pair': (c_int, [c_int, POINTER(LstmBwdProblem), POINTER(LstmBwdProblem), P]),
    'a4nt_lstm_check_timeout': (c_int, [P, c_int, P]),
    'a4nt_lstm_dec_step': (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P]),
    'a4nt_lstm_dec_step_merge': (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, c_int, P, c_float, c_int, c_int, P, P,
                                         P, P, P, P, c_longlong, P]),
    'a4nt_gen_loop_fwd': (c_int, [POINTER(GenFwdArgs), c_int, c_int, P]),
    'a4nt_gen_loop_bwd': (c_int, [POINTER(GenBwdArgs), P]),
    'a4nt_gen_persist_ok': (c_int, [c_int, c_int, c_int]),
    'a4nt_gen_persist_xch_floats': (c_longlong, [c_int, c_int]),
    'a4nt_gen_persist_sync_words': (c_int, []),
    'a4nt_gen_persist_fwd': (c_int, [POINTER(GenFwdArgs), c_int, c_int, P, P, P]),
    'a4nt_bce_logits_fwd': (c_int, [c_int, c_int, c_int, P, c_float, P, P, P]),
    'a4nt_bce_logits_bwd': (c_int, [c_int, c_int, c_int, P, c_float, P, P, P, P]),
    'a4nt_mt_chunk_elems': (c_int, []),
    'a4nt_mt_sumsq': (c_int, [c_int, P, P, P, P]),
    'a4nt_mt_clip_coef': (c_int, [c_int, P, c_float, P, P]),
    'a4nt_mt_scale': (c_int, [c_int, P, P, P, P]),
    'a4nt_mt_rmsprop': (c_int, [c_int, P, P, c_float, c_float, c_float, c_float, P]),
    'a4nt_lstm_cell_bwd': (c_int, [c_int, c_int, P, P, P, P, c_int, c_longlong, P, c_int, P, P, P]),
    'a4nt_gumbel_fwd': (c_int, [c_int, c_int, P, c_int, P, c_int, c_ulonglong, c_uint, c_float, c_float, c_int, c_int,
                                P, c_int, P, P, P, P, P, P, c_longlong, P, c_int, P]),
    'a4nt_gumbel_bwd': (c_int, [c_int, c_int, P, c_int, P, c_int, P, c_float, c_float, P, c_int, P]),
    'a4nt_proj_gumbel_packed_floats': (c_longlong, [c_int, c_int]),
    'a4nt_proj_gumbel_pack': (c_int, [c_int, c_int, P, c_int, P, P]),
    'a4nt_proj_gumbel_part_floats': (c_longlong, [c_int, c_int]),
    'a4nt_proj_gumbel_fwd': (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_float, c_float, c_int, c_int, P, c_int,
                                     c_ulonglong, c_uint, P, c_int, c_int, c_int, P, P, P, P, P, P, P, c_longlong, P]),
    'a4nt_proj_gumbel_finalize': (c_int, [c_int, c_int, P, c_float, c_int, c_int, P, P, P, P, P, P, c_longlong, P]),
    'a4nt_nll_rows': (c_int, [c_int, c_int, P, c_int, P, c_float, P, c_int, P]),
    'a4nt_sgemm_nll_parts': (c_int, [c_int]),
    'a4nt_sgemm_nll': (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, P, P]),
    'a4nt_nll_grad_rows': (c_int, [c_int, c_int, P, c_int, P, P, c_int, c_float, P, c_int, P]),
    'a4nt_embed_fwd': (c_int, [c_longlong, c_int, c_int, P, P, P, P, c_int, P]),
    'a4nt_embed_bwd': (c_int, [c_longlong, c_int, c_int, P, P, P, c_int, P, c_longlong, P]),
    'a4nt_colsum_ws_floats': (c_longlong, [c_int, c_int]),
    'a4nt_colsum': (c_int, [c_int, c_int, P, c_int, P, c_int, P, c_longlong, P]),
    'a4nt_sgemm_tn_colsum': (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, c_int, c_longlong, P, P]),
    'a4nt_slab_reduce': (c_int, [c_int, c_longlong, c_int, c_int, P, c_int, c_int, P, c_int, P, c_int, P]),
    'a4nt_slab_reduce2': (c_int, [c_int, c_longlong, c_int, c_int, P, c_int, P, c_int, c_int, P, c_int, P, c_int, P]),
    'a4nt_conv1d_tm': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, c_int, c_float, P, c_int, c_int,
                               c_int, c_int, c_int, P, P]),
    'a4nt_maxkey_init': (c_int, [c_int, c_int, P, c_int, P]),
    'a4nt_maxkey_decode': (c_int, [c_int, c_int, P, P, c_int, P, P]),
    'a4nt_maxtime_scatter_bwd': (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_float, P, c_int, P]),
    'a4nt_lrelu_maxtime': (c_int, [c_int, c_int, c_int, P, c_int, c_float, c_int, P, c_int, P, P]),
    'a4nt_lrelu_maxtime_bwd': (c_int, [c_int, c_int, c_int, P, c_int, c_float, P, c_int, P, P]),
}
EXPORTS = sorted(_SIGS)

_lib = None


class A4NTError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise A4NTError('%s not found: build it with a4nt-author-masking_amd/csrc/build.sh (or __graft_entry__.build()); '
                        'there is no CPU fallback for the HIP path' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().a4nt_last_error().decode('utf-8', 'replace')
        raise A4NTError('%s failed (code %d): %s' % (what or 'a4nt call', rc, msg))


_inited = set()


def init_device(index):
    if index in _inited:
        return
    ncu = c_int(0)
    check(load().a4nt_init(int(index), ctypes.byref(ncu)), 'a4nt_init')
    _inited.add(index)
