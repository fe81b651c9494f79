"""GPU: a4nt_lstm_seq_fwd_noalloc (the entry point that allocates nothing: hand-off buffer and counters in a static
device array of the library) against a4nt_lstm_seq_fwd with a caller-provided workspace -- same kernels, so every
output must be bit-identical -- on two calls in a row, the second on a smaller input (the static array keeps the
first call's contents).  Called through the C ABI (ctypes), not through the Python operators.  Collected last
(file name): `pytest -x` must not lose other tests behind a failure here."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _run(lib, noalloc, T, B, H, seed, ragged, own_word=True):
    from a4nt_amd import _lib
    g = torch.Generator().manual_seed(seed)
    dev = torch.device('cuda:0')
    gx = (torch.randn(T, B, 4 * H, generator=g) * 0.5).to(dev)
    gc = (torch.randn(B, 4 * H, generator=g) * 0.1).to(dev)
    whh = (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dev)
    h0 = (torch.randn(B, H, generator=g) * 0.3).to(dev)
    c0 = (torch.randn(B, H, generator=g) * 0.3).to(dev)
    lens = None
    if ragged:
        lens = torch.randint(1, T + 1, (B,), generator=g).int()
        lens[0] = T
        lens = lens.to(dev)
    hs = torch.full((T, B, H), float('nan'), device=dev)
    gates = torch.full((T, B, 4 * H), float('nan'), device=dev)
    cs = torch.full((T, B, H), float('nan'), device=dev)
    hN = torch.full((B, H), float('nan'), device=dev)
    cN = torch.full((B, H), float('nan'), device=dev)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if noalloc:
        tmo = torch.zeros(1, dtype=torch.int32, device=dev)
        call = lambda: lib.a4nt_lstm_seq_fwd_noalloc(T, B, H, _p(gx), _p(gc), _p(whh), _p(h0), _p(c0), _p(lens), _p(hs),
                                                     _p(gates), _p(cs), _p(hN), _p(cN), _p(tmo) if own_word else None, s)
        _lib.check(call(), 'a4nt_lstm_seq_fwd_noalloc')      # warm-up: lazy code-object loading is not an allocation of the call
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        free_before = torch.cuda.mem_get_info()[0]           # driver level: sees a hipMalloc / hipFree torch's counter cannot
        _lib.check(call(), 'a4nt_lstm_seq_fwd_noalloc')
        assert torch.cuda.memory_allocated() == before
        assert torch.cuda.mem_get_info()[0] == free_before, 'the call changed the device\'s free memory'
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
    else:
        nf = lib.a4nt_lstm_xch_floats(B, H, 0)
        nw = lib.a4nt_lstm_sync_words(B, H)
        assert nf > 0 and nw > 0
        xch = torch.zeros(nf, device=dev)
        sync = torch.zeros(nw, dtype=torch.int32, device=dev)
        tmo = sync[nw - 32:nw - 31]
        rc = lib.a4nt_lstm_seq_fwd(T, B, H, _p(gx), _p(gc), _p(whh), _p(h0), _p(c0), _p(lens), _p(hs), _p(gates), _p(cs),
                                   _p(hN), _p(cN), _p(xch), _p(sync), s)
        _lib.check(rc, 'a4nt_lstm_seq_fwd')
    torch.cuda.synchronize()
    assert int(tmo.item()) == 0, 'persistent sweep timed out'
    return [t.cpu() for t in (hs, gates, cs, hN, cN)], lens


@pytest.mark.timeout(300)
def test_noalloc_forward_sweep_matches_workspace_entry_point():
    from a4nt_amd import _lib
    lib = _lib.load()
    torch.cuda.set_device(0)
    # first a large problem (fills the static array: 128 rows x 512 units), then smaller ones that reuse it
    # (the third case passes no timeout word: the library's internal word, cleared per call, is used)
    for (T, B, H, ragged) in [(9, 128, 512, True), (5, 20, 128, True), (4, 3, 64, False), (6, 100, 1024, True)]:
        got, lens = _run(lib, True, T, B, H, 100 + B, ragged, own_word=(B != 3))
        want, _ = _run(lib, False, T, B, H, 100 + B, ragged)
        for name, a, b in zip(('hs', 'gates', 'cs', 'hN', 'cN'), got, want):
            if name in ('gates', 'cs') and lens is not None:
                # steps past the end of the longest row of a group are never written by either entry point
                t = torch.arange(T).view(T, 1, 1)
                m = (t < lens.cpu().view(1, B, 1)).expand_as(a)
                a, b = torch.where(m, a, torch.zeros_like(a)), torch.where(m, b, torch.zeros_like(b))
            assert not torch.isnan(a).any(), name
            assert torch.equal(a, b), '%s differs (T=%d B=%d H=%d)' % (name, T, B, H)


def test_noalloc_rejects_oversize():
    from a4nt_amd import _lib
    lib = _lib.load()
    torch.cuda.set_device(0)
    d = torch.zeros(16, device='cuda:0')
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # 4096 rows at H = 512 exceed any single-sweep geometry, and with it the static array
    rc = lib.a4nt_lstm_seq_fwd_noalloc(2, 4096, 512, _p(d), None, _p(d), None, None, None, _p(d), None, None, None, None,
                                       None, s)
    assert rc == 1
    assert b'too large' in lib.a4nt_last_error() or b'static scratch' in lib.a4nt_last_error()
    rc = lib.a4nt_lstm_seq_fwd_noalloc(2, 16, 100, _p(d), None, _p(d), None, None, None, _p(d), None, None, None, None,
                                       None, s)
    assert rc == 1
