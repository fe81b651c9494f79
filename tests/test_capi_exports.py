"""CPU: the C-ABI library builds/loads and exports every symbol include/a4nt_hip.h declares, and the
ctypes binding table matches the header (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import a4nt_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'a4nt_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(a4nt_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(a4nt_amd.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), 'liba4nt_hip.so does not export %s' % n


def test_binding_table_matches_header():
    assert sorted(a4nt_amd.EXPORTS) == _declared()


def test_argument_errors_without_gpu():
    lib = a4nt_amd.load_library()
    mt, g = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.a4nt_lstm_geometry(128, 512, ctypes.byref(mt), ctypes.byref(g)) == 0
    assert (mt.value, g.value) == (1, 8)
    assert lib.a4nt_lstm_geometry(256, 512, ctypes.byref(mt), ctypes.byref(g)) == 0
    assert (mt.value, g.value) == (1, 16)          # two workgroups per CU
    assert lib.a4nt_lstm_set_max_workgroups(256) == 0
    assert lib.a4nt_lstm_geometry(256, 512, ctypes.byref(mt), ctypes.byref(g)) == 0
    assert (mt.value, g.value) == (2, 8)
    assert lib.a4nt_lstm_set_max_workgroups(512) == 0
    assert lib.a4nt_lstm_geometry(512, 512, ctypes.byref(mt), ctypes.byref(g)) == 0
    assert (mt.value, g.value) == (4, 8)
    assert lib.a4nt_lstm_geometry(5, 64, ctypes.byref(mt), ctypes.byref(g)) == 0
    assert (mt.value, g.value) == (1, 1)
    assert lib.a4nt_lstm_geometry(8, 100, ctypes.byref(mt), ctypes.byref(g)) == 1      # H not instantiated
    assert b'H must be 64, 128, 256, 512 or 1024' in lib.a4nt_last_error()
    assert lib.a4nt_lstm_geometry(256, 1024, ctypes.byref(mt), ctypes.byref(g)) == 0
    assert (mt.value, g.value) == (4, 4)
    assert lib.a4nt_lstm_geometry(4096, 512, ctypes.byref(mt), ctypes.byref(g)) == 1   # batch too large
    assert lib.a4nt_version() == 100
    assert lib.a4nt_lstm_xch_floats(128, 512, 0) == 2 * 8 * 32 * 256
    assert lib.a4nt_lstm_xch_floats(128, 512, 1) == 2 * 8 * 32 * 256 * 32     # [parity][group][consumer][producer] blocks
    assert lib.a4nt_lstm_xch_floats(100, 512, 0) == 2 * 128 * 512             # rows padded to 64: any rows-per-group fits
    assert lib.a4nt_lstm_sync_words(128, 512) == 32 * 8 + 32


def test_no_cpu_fallback():
    """The HIP operators refuse CPU tensors instead of silently computing elsewhere."""
    import pytest
    import torch
    from a4nt_amd import ops
    with pytest.raises(a4nt_amd.A4NTError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
