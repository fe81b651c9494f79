"""Host-side operators over the C-ABI kernels (include/a4nt_hip.h).

Everything here allocates with torch, passes raw device pointers + the current HIP stream to
liba4nt_hip.so and wires the results into torch.autograd.  There is no CPU / eager fallback: a
tensor that is not a float32 CUDA tensor is an error.
"""
import contextlib
import os
import ctypes
import threading

import torch
from torch.autograd import Function

from . import _lib
from ._lib import check, A4NTError


# ----------------------------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------------------------
def _p(t):
    return None if t is None else t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


def _f32(t, what):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise A4NTError('%s must be a float32 CUDA tensor (got %s on %s); the HIP path has no CPU fallback'
                        % (what, t.dtype, t.device))
    return t


# Optional per-kernel timing for bench.py's roofline line: TIMER = {'name': <kernel label>, 'events': []}.
# Labels: 'sgemm_<nn|nt|tn>_<128|64>' (one per sgemm_kernel instantiation), 'lstm_seq_fwd',
# 'lstm_seq_bwd', 'proj_gumbel'.
# Events are recorded on torch's current stream, which is the stream every kernel here is launched on.
TIMER = None


class _Timed(object):
    def __init__(self, name, flops):
        self.on = TIMER is not None and TIMER['name'] == name
        self.flops = flops

    def __enter__(self):
        if self.on:
            pool = TIMER.get('pool')
            if pool is not None:
                # events made before the timed region (creating them inside it costs host time per launch); when the
                # pool runs dry the remaining launches go untimed
                if len(pool) < 2:
                    self.on = False
                    return self
                self.e0, self.e1 = pool.pop(), pool.pop()
            else:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            TIMER['events'].append((self.e0, self.e1, self.flops))
        return False


def make_timer(name, max_launches=None):
    """TIMER dictionary for kernel label `name`; max_launches pre-creates that many event pairs."""
    t = {'name': name, 'events': []}
    if max_launches:
        t['pool'] = [torch.cuda.Event(enable_timing=True) for _ in range(2 * int(max_launches))]
    return t


def _rowmajor(t):
    """2-D view with unit inner stride (leading dimension = stride(0))."""
    assert t.dim() == 2
    if (t.shape[1] > 1 and t.stride(1) != 1) or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _ld(t):
    return max(int(t.stride(0)), int(t.shape[1])) if t.shape[0] > 1 else int(max(t.shape[1], 1))


NT_ROWS = True      # route few-row NT products with K <= 512 to a4nt_sgemm_nt_rows (benchmarks switch it off)


def gemm(A, B, transA=False, transB=False, bias=None, out=None, beta=0, ksplit=1):
    """out[M,N] (+)= op(A) op(B) (+ bias).  ksplit>1 returns the [ksplit,M,N] partial slabs."""
    lib = _lib.load()
    A = _rowmajor(_f32(A, 'gemm A'))
    B = _rowmajor(_f32(B, 'gemm B'))
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    N, Kb = (B.shape[0], B.shape[1]) if transB else (B.shape[1], B.shape[0])
    if K != Kb:
        raise A4NTError('gemm: inner dimensions differ (%d vs %d)' % (K, Kb))
    if out is None:
        out = torch.empty((ksplit, M, N) if ksplit > 1 else (M, N), device=A.device, dtype=torch.float32)
        ldc = N
    else:
        _f32(out, 'gemm out')
        o2 = out if out.dim() == 2 else out.view(-1, out.shape[-1])
        assert o2.stride(1) == 1 or o2.shape[1] == 1
        ldc = _ld(o2)
    if M == 0 or N == 0:
        return out
    if K == 0:
        if not beta:
            out.zero_()
        return out
    if not transA and not transB and ksplit == 1 and M <= 256 and N >= 2048 and K >= 64:
        # few rows x wide [K,N] weight (one decoder step's projection): the LDS-free per-wave kernel is faster
        label = 'skinny_nn' if TIMER is not None else None
        with _Timed(label, 2.0 * M * N * K):
            check(lib.a4nt_sgemm_skinny(0, M, N, K, _p(A), _ld(A), _p(B), _ld(B), _p(out), ldc, _p(bias), int(beta), 1, 0,
                                        _s()), 'a4nt_sgemm_skinny')
        return out
    if (transB and not transA and ksplit == 1 and bias is None and 96 <= M <= 256 and N >= 4096 and NT_ROWS
            and lib.a4nt_sgemm_nt_rows_ok(M, N, K, _ld(A), _ld(B)) and (A.data_ptr() | B.data_ptr()) % 16 == 0):
        # few rows x [N,K] weight with K <= 512 (one decoder step's d(sample) += d(emb) E^T): A-stationary kernel
        with _Timed('sgemm_nt_rows' if TIMER is not None else None, 2.0 * M * N * K):
            check(lib.a4nt_sgemm_nt_rows(M, N, K, _p(A), _ld(A), _p(B), _ld(B), _p(out), ldc, int(beta), _s()),
                  'a4nt_sgemm_nt_rows')
        return out
    label = None
    if TIMER is not None:
        label = 'sgemm_%s_%d' % ('tn' if transA else ('nt' if transB else 'nn'), lib.a4nt_sgemm_tile(M, N, int(ksplit)))
    with _Timed(label, 2.0 * M * N * K):
        check(lib.a4nt_sgemm(int(transA), int(transB), M, N, K, _p(A), _ld(A), _p(B), _ld(B), _p(out), ldc,
                             _p(bias), int(beta), int(ksplit), M * ldc if ksplit > 1 else 0, _s()), 'a4nt_sgemm')
    return out


def gemm_nt_rows(A, B, out=None, beta=0):
    """out (+)= A @ B^T through a4nt_sgemm_nt_rows (few rows, K <= 512; raises when the shape does not qualify)."""
    lib = _lib.load()
    A = _rowmajor(_f32(A, 'gemm_nt_rows A'))
    B = _rowmajor(_f32(B, 'gemm_nt_rows B'))
    M, K = A.shape
    N = B.shape[0]
    if B.shape[1] != K:
        raise A4NTError('gemm_nt_rows: inner dimensions differ (%d vs %d)' % (K, B.shape[1]))
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
        beta = 0
    assert out.dim() == 2 and (out.stride(1) == 1 or out.shape[1] == 1)
    check(lib.a4nt_sgemm_nt_rows(M, N, K, _p(A), _ld(A), _p(B), _ld(B), _p(out), _ld(out), int(beta), _s()),
          'a4nt_sgemm_nt_rows')
    return out


def skinny_ksplit(M, N, K):
    return int(_lib.load().a4nt_sgemm_skinny_ksplit(int(M), int(N), int(K)))


def gemm_skinny(A, B, transB=False, bias=None, out=None, beta=0, ksplit=1):
    """out (+)= A @ op(B) (+ bias) through a4nt_sgemm_skinny (few rows: one decoder step's batch).
    ksplit > 1: `out` must be [ksplit, M, N] slabs (or None: allocated)."""
    lib = _lib.load()
    A = _rowmajor(_f32(A, 'gemm_skinny A'))
    B = _rowmajor(_f32(B, 'gemm_skinny B'))
    M, K = A.shape
    N, Kb = (B.shape[0], B.shape[1]) if transB else (B.shape[1], B.shape[0])
    if K != Kb:
        raise A4NTError('gemm_skinny: inner dimensions differ (%d vs %d)' % (K, Kb))
    if out is None:
        out = torch.empty((ksplit, M, N) if ksplit > 1 else (M, N), device=A.device, dtype=torch.float32)
        beta = 0
    if ksplit > 1:
        assert out.dim() == 3 and out.shape[0] == ksplit and out.is_contiguous() and not beta
        ldc, slab = N, M * N
    else:
        assert out.dim() == 2 and out.stride(1) == 1
        ldc, slab = out.stride(0), 0
    label = ('skinny_nt' if transB else 'skinny_nn') if TIMER is not None else None
    with _Timed(label, 2.0 * M * N * K):
        check(lib.a4nt_sgemm_skinny(int(transB), M, N, K, _p(A), _ld(A), _p(B), _ld(B), _p(out), ldc, _p(bias), int(beta),
                                    int(ksplit), slab, _s()), 'a4nt_sgemm_skinny')
    return out


def padded_rows(N, V, device, lead=()):
    """[*lead, N, V] fp32 view whose rows start 16-byte aligned (leading dimension rounded up to 16)."""
    Vp = (V + 15) // 16 * 16
    return torch.empty(tuple(lead) + (N, Vp), device=device, dtype=torch.float32)[..., :V]


def slab_reduce(slabs, c0=0, ncols=None, add=None):
    """sum over the leading (split-K) axis of slabs[S,R,C_total][:, :, c0:c0+ncols] (+ add)."""
    lib = _lib.load()
    S, R, Ct = slabs.shape
    C = Ct - c0 if ncols is None else ncols
    out = torch.empty((R, C), device=slabs.device, dtype=torch.float32)
    check(lib.a4nt_slab_reduce(S, R * Ct, R, C, _p(slabs), Ct, c0, _p(add), C if add is not None else 0, _p(out), C,
                               _s()), 'a4nt_slab_reduce')
    return out


def _auto_ksplit(M, N, K):
    """Split-K factor for gemm_auto.

    Few output tiles: split K so that the 256 CUs have work (a4nt_sgemm picks the 128x128 kernel once
    tiles*ksplit >= 200, the latency-tolerant 64x64 one otherwise).  Many tiles: the 128x128 kernel runs
    at ~0.48 TFLOP/s per CU (the fp32 MFMA rate at the clock the chip holds), so the launch takes
    ceil(tiles*ks/256) rounds of one K/ks-deep tile each; a tile count just above a multiple of 256
    (316 tiles for the [512 x 10001] weight gradient) wastes most of a round, and splitting K trades
    that for the slab write + reduce traffic.  Pick the cheapest under that model."""
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    if K < 1024:
        return 1
    if t128 < 32:
        t64 = ((M + 63) // 64) * ((N + 63) // 64)
        return int(max(1, min(512 // max(t64, 1), K // 256, 32)))
    best, best_t = 1, None
    for ks in range(1, int(min(16, K // 512)) + 1):
        rounds = (t128 * ks + 255) // 256
        t = rounds * (0.06875 * ((K + ks - 1) // ks) + 4.0)                 # us
        # one workgroup per CU is one wave per SIMD: its load / barrier waits show (measured on the 64-tile
        # [2048 x 512] weight gradients: 280 us with 256 workgroups, 240 us with 512 or 768)
        wgs_per_cu = t128 * ks / 256.0
        t /= 0.85 if wgs_per_cu < 1.5 else 1.0
        if ks > 1:
            # slab write + reduce pass: slabs that fit the Infinity Cache cost about a third of HBM-sized ones
            # ([2048 x 512] x 16 slabs: +3 us over 8 slabs; [512 x 10001]: +42 us from 4 to 8 slabs)
            bw = 3.5e6 if ks * M * N * 4 > 96e6 else 1.0e7
            t += (2 * ks + 1) * M * N * 4 / bw + 6.0
        if best_t is None or t < 0.97 * best_t:
            best, best_t = ks, t
    return best


def gemm_auto(A, B, transA=False, transB=False, bias=None):
    """gemm that splits K (deterministic slab reduction) when the output has too few tiles to fill the chip."""
    A2, B2 = _rowmajor(A), _rowmajor(B)
    M, K = (A2.shape[1], A2.shape[0]) if transA else (A2.shape[0], A2.shape[1])
    N = B2.shape[0] if transB else B2.shape[1]
    ks = _auto_ksplit(M, N, K)
    if ks <= 1:
        return gemm(A2, B2, transA, transB, bias=bias)
    slabs = gemm(A2, B2, transA, transB, bias=bias, ksplit=ks)
    return slab_reduce(slabs)


def colsum(A, out=None, accumulate=False):
    lib = _lib.load()
    A = _rowmajor(_f32(A, 'colsum A'))
    M, N = A.shape
    if out is None:
        out = torch.empty(N, device=A.device, dtype=torch.float32)
        accumulate = False
    nws = int(lib.a4nt_colsum_ws_floats(M, N))
    ws = torch.empty(nws, device=A.device, dtype=torch.float32) if nws else None
    check(lib.a4nt_colsum(M, N, _p(A), _ld(A), _p(out), int(accumulate), _p(ws), nws, _s()), 'a4nt_colsum')
    return out


def gemm_tn_colsum(A, B):
    """(A^T @ B, column sums of B) for A [K,M], B [K,N]: weight and bias gradient of y = x W + b from x and dy in one
    pass over dy (a4nt_sgemm_tn_colsum; split-K as gemm_auto picks it).  Few-tile outputs take the two separate calls."""
    lib = _lib.load()
    A = _rowmajor(_f32(A, 'gemm_tn_colsum A'))
    B = _rowmajor(_f32(B, 'gemm_tn_colsum B'))
    K, M = A.shape
    N = B.shape[1]
    if B.shape[0] != K:
        raise A4NTError('gemm_tn_colsum: inner dimensions differ (%d vs %d)' % (K, B.shape[0]))
    ks = _auto_ksplit(M, N, K)
    if not TN_COLSUM or K == 0 or lib.a4nt_sgemm_tile(M, N, ks) != 128:
        return gemm_auto(A, B, transA=True), colsum(B)
    out = torch.empty((ks, M, N) if ks > 1 else (M, N), device=A.device, dtype=torch.float32)
    colpart = torch.empty((ks, N), device=A.device, dtype=torch.float32)
    with _Timed('sgemm_tn_128_colsum' if TIMER is not None else None, 2.0 * M * N * K):
        check(lib.a4nt_sgemm_tn_colsum(M, N, K, _p(A), _ld(A), _p(B), _ld(B), _p(out), N, ks, M * N if ks > 1 else 0,
                                       _p(colpart), _s()), 'a4nt_sgemm_tn_colsum')
    return (slab_reduce(out) if ks > 1 else out), (colsum(colpart) if ks > 1 else colpart[0])


TN_COLSUM = os.environ.get('A4NT_TN_COLSUM', '1') != '0'   # 0: gemm_tn_colsum runs the product and a4nt_colsum separately (A/B runs)


def embed_fwd(ids, W, scale=None):
    lib = _lib.load()
    ids = ids.contiguous()
    assert ids.dtype == torch.int64 and ids.is_cuda
    N, (V, E) = ids.numel(), W.shape
    out = torch.empty(tuple(ids.shape) + (E,), device=W.device, dtype=torch.float32)
    check(lib.a4nt_embed_fwd(N, E, V, _p(ids), _p(scale), _p(_f32(W, 'embedding').contiguous()), _p(out), E, _s()),
          'a4nt_embed_fwd')
    return out


def embed_bwd(ids, dout, dW, scale=None, padding_idx=-1):
    lib = _lib.load()
    ids = ids.contiguous()
    dout = dout.contiguous()
    V, E = dW.shape
    check(lib.a4nt_embed_bwd(ids.numel(), E, V, _p(ids), _p(scale), _p(dout), E, _p(dW), int(padding_idx), _s()),
          'a4nt_embed_bwd')
    return dW


# ----------------------------------------------------------------------------------------------
# persistent-sweep workspaces (one per device / shape; calls on one stream are ordered)
# ----------------------------------------------------------------------------------------------
_ws = {}
_sync_blocks = []


def _ensure_init():
    """a4nt_init on the current device before any geometry question: the budgets depend on its occupancy probe."""
    if torch.cuda.is_available():
        _lib.init_device(torch.cuda.current_device())


def lstm_geometry(B, H):
    lib = _lib.load()
    _ensure_init()
    mt, g = ctypes.c_int(0), ctypes.c_int(0)
    check(lib.a4nt_lstm_geometry(B, H, ctypes.byref(mt), ctypes.byref(g)), 'a4nt_lstm_geometry')
    return mt.value, g.value


def _lstm_ws(dev, B, H, backward):
    """Hand-off buffer + sync block of a sweep.  Keyed on the geometry in force NOW (rows per group, groups): the
    library derives (mt, G) at launch time from the workgroup budget, which lstm_set_max_workgroups and
    BWD_MAX_WGS change, and the sync block's size and the place of its timeout word (32*G) follow G."""
    _lib.init_device(dev.index if dev.index is not None else torch.cuda.current_device())
    mt, g = lstm_geometry(B, H)
    key = (dev.index, B, H, backward, mt, g)
    if key not in _ws:
        lib = _lib.load()
        nf = lib.a4nt_lstm_xch_floats(B, H, int(backward))
        nw = lib.a4nt_lstm_sync_words(B, H)
        if nf < 0 or nw < 0:
            check(1, 'a4nt_lstm workspace')
        assert nw == 32 * g + 32
        xch = torch.zeros(nf, device=dev, dtype=torch.float32)
        sync = torch.zeros(nw, device=dev, dtype=torch.int32)
        _ws[key] = (xch, sync, g)
        _sync_blocks.append((sync, g))
    return _ws[key]


def lstm_rows_fit(B, H):
    """True when a batch of B rows fits one persistent sweep at hidden size H."""
    mt, g = ctypes.c_int(0), ctypes.c_int(0)
    _ensure_init()
    return _lib.load().a4nt_lstm_geometry(int(B), int(H), ctypes.byref(mt), ctypes.byref(g)) == 0


def lstm_set_max_workgroups(n):
    """256: one persistent-sweep workgroup per CU (leaves register space for kernels of other streams,
    e.g. RCCL under data parallelism); 512 (default): two per CU where the geometry allows it."""
    check(_lib.load().a4nt_lstm_set_max_workgroups(int(n)), 'a4nt_lstm_set_max_workgroups')
    _MAX_WGS[_cur_dev()] = int(n)      # the library keeps this setting per device (the calling thread's current one)


def _cur_dev():
    return torch.cuda.current_device() if torch.cuda.is_available() else -1


_MAX_WGS = {}      # device index -> budget set through lstm_set_max_workgroups (absent: the library's default, 512)
# Budget of the BACKWARD sweeps alone (None: the same as the forward sweeps').  Data parallelism: the gradient
# all-reduces run under the backward passes, so those sweeps leave room for RCCL's kernels (256), while the forward
# sweeps -- no collective is in flight then: every exchange is waited for before the optimizer step -- keep two
# workgroups per CU.
BWD_MAX_WGS = None
# BWD_MAX_WGS applies only to sweeps that are QUEUED while a gradient all-reduce is in flight (issued, not yet waited
# for: adv_step.GradAllReducer counts them in COLLECTIVES_IN_FLIGHT).  A collective starts behind everything queued
# on the compute stream before it was issued and is waited for before the optimizer step, so only the sweeps queued
# in between can share the chip with its kernels: the BPTT sweeps in front of the first complete bucket keep two
# workgroups per CU.  False: every backward sweep takes BWD_MAX_WGS.
BWD_BUDGET_DYNAMIC = os.environ.get('A4NT_BWD_BUDGET_DYNAMIC', '1') != '0'
COLLECTIVES_IN_FLIGHT = 0
BWD_BUDGET_STATS = {'switched': 0, 'kept': 0}       # backward launches under BWD_MAX_WGS: with / without the switch


@contextlib.contextmanager
def _bwd_budget():
    """Backward sweeps under BWD_MAX_WGS: the budget of the device this thread runs on is switched for the duration
    of the launches and restored to THAT device's own setting.  One host thread per device at a time (the library's
    budget is a per-device global: two threads launching sweeps on one device while one of them is inside this
    context would see each other's budget)."""
    n = BWD_MAX_WGS
    mine = _MAX_WGS.get(_cur_dev(), 512)
    if n is None or int(n) == mine:
        yield
        return
    if BWD_BUDGET_DYNAMIC and COLLECTIVES_IN_FLIGHT <= 0:
        BWD_BUDGET_STATS['kept'] += 1
        yield
        return
    BWD_BUDGET_STATS['switched'] += 1
    lib = _lib.load()
    check(lib.a4nt_lstm_set_max_workgroups(int(n)), 'a4nt_lstm_set_max_workgroups')
    try:
        yield
    finally:
        check(lib.a4nt_lstm_set_max_workgroups(mine), 'a4nt_lstm_set_max_workgroups')


def check_timeouts():
    """Raise if any persistent sweep since the last call gave up on an inter-workgroup wait (host sync).  The
    timeout word of a sync block is sticky (launches clear the counters only); it is cleared here."""
    for sync, g in _sync_blocks:
        if int(sync[32 * g].item()) != 0:
            sync[32 * g] = 0
            raise A4NTError('persistent LSTM sweep timed out waiting for a peer workgroup '
                            '(workgroups not co-resident?); results of that sweep are invalid')


def lstm_seq_fwd(gx, w_hh, lens_dev, h0=None, c0=None, gc=None, save=True):
    """gx: [T,B,4H] pre-activations (overwritten with the activations).  Returns hs_all [T+1,B,H]
    (row 0 = h0), cs [T,B,H] or None, hN, cN."""
    lib = _lib.load()
    T, B, H4 = gx.shape
    H = H4 // 4
    dev = gx.device
    xch, sync, _ = _lstm_ws(dev, B, H, False)
    hs_all = torch.empty((T + 1, B, H), device=dev, dtype=torch.float32)
    if h0 is None:
        hs_all[0].zero_()
    else:
        hs_all[0].copy_(h0)
    cs = torch.empty((T, B, H), device=dev, dtype=torch.float32) if save else None
    hN = torch.empty((B, H), device=dev, dtype=torch.float32)
    cN = torch.empty((B, H), device=dev, dtype=torch.float32)
    with _Timed('lstm_seq_fwd', 8.0 * T * B * H * H):
        check(lib.a4nt_lstm_seq_fwd(T, B, H, _p(gx), _p(gc), _p(w_hh), _p(h0), _p(c0), _p(lens_dev), _p(hs_all[1:]),
                                    _p(gx) if save else None, _p(cs), _p(hN), _p(cN), _p(xch), _p(sync), _s()),
              'a4nt_lstm_seq_fwd')
    return hs_all, cs, hN, cN


def lstm_seq_bwd(w_hh, gates, cs, c0, lens_dev, dhs, dhN, dcN, need_h0):
    lib = _lib.load()
    T, B, H4 = gates.shape
    H = H4 // 4
    dev = gates.device
    dgates = torch.empty_like(gates)
    dh0 = torch.empty((B, H), device=dev, dtype=torch.float32) if need_h0 else None
    dc0 = torch.empty((B, H), device=dev, dtype=torch.float32) if need_h0 else None
    with _bwd_budget():
        xch, sync, g = _lstm_ws(dev, B, H, True)
        # H >= 256: the sweep leaves the bias gradient's per-row-group partial sums (no pass over dgates for them)
        dbp = torch.empty((g, H4), device=dev, dtype=torch.float32) if H >= 256 else None
        with _Timed('lstm_seq_bwd', 8.0 * T * B * H * H):
            check(lib.a4nt_lstm_seq_bwd_bias(T, B, H, _p(w_hh), _p(gates), _p(cs), _p(c0), _p(lens_dev), _p(dhs), _p(dhN),
                                             _p(dcN), _p(dgates), _p(dh0), _p(dc0), _p(dbp), _p(xch), _p(sync), _s()),
                  'a4nt_lstm_seq_bwd_bias')
    dgates._a4nt_dbias = dbp
    return dgates, dh0, dc0


def lstm_bias_grad(dgates):
    """d(b_ih) = d(b_hh) = column sums of dgates [T,B,4H]: from the partials the backward sweep left, else a pass."""
    part = getattr(dgates, '_a4nt_dbias', None)
    if part is not None:
        return part.sum(0)
    return colsum(dgates.view(-1, dgates.shape[-1]))


def lstm_pair_geometry(B0, B1, H):
    """(mt, g0, g1) when two sweeps of B0 and B1 rows fit one launch at hidden size H, else None."""
    mt, g0, g1 = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _ensure_init()
    if _lib.load().a4nt_lstm_pair_geometry(int(B0), int(B1), int(H), ctypes.byref(mt), ctypes.byref(g0),
                                           ctypes.byref(g1)) != 0:
        return None
    return mt.value, g0.value, g1.value


def _lstm_pair_ws(dev, B, H, mt, backward, slot):
    key = (dev.index, B, H, backward, 'pair', mt, slot)
    if key not in _ws:
        lib = _lib.load()
        _lib.init_device(dev.index if dev.index is not None else torch.cuda.current_device())
        nf = lib.a4nt_lstm_pair_xch_floats(B, H, mt, int(backward))
        nw = lib.a4nt_lstm_pair_sync_words(B, mt)
        if nf < 0 or nw < 0:
            check(1, 'a4nt_lstm pair workspace')
        xch = torch.zeros(nf, device=dev, dtype=torch.float32)
        sync = torch.zeros(nw, device=dev, dtype=torch.int32)
        g = (nw - 32) // 32
        _ws[key] = (xch, sync, g)
        _sync_blocks.append((sync, g))
    return _ws[key]


def lstm_seq_fwd_pair(reqs, save=True):
    """Two forward sweeps in one launch.  reqs: two tuples (gx, w_hh, lens_dev, h0, c0, gc) as for lstm_seq_fwd
    (same hidden size; lstm_pair_geometry must accept the two batch sizes).  Returns two lstm_seq_fwd results."""
    lib = _lib.load()
    H = reqs[0][0].shape[2] // 4
    dev = reqs[0][0].device
    mt = lstm_pair_geometry(reqs[0][0].shape[1], reqs[1][0].shape[1], H)[0]
    probs, outs, keep = [], [], []
    flops = 0.0
    for slot, (gx, w_hh, lens_dev, h0, c0, gc) in enumerate(reqs):
        T, B, _ = gx.shape
        xch, sync, _g = _lstm_pair_ws(dev, B, H, mt, False, slot)
        hs_all = torch.empty((T + 1, B, H), device=dev, dtype=torch.float32)
        if h0 is None:
            hs_all[0].zero_()
        else:
            hs_all[0].copy_(h0)
        cs = torch.empty((T, B, H), device=dev, dtype=torch.float32) if save else None
        hN = torch.empty((B, H), device=dev, dtype=torch.float32)
        cN = torch.empty((B, H), device=dev, dtype=torch.float32)
        probs.append(_lib.LstmFwdProblem(T, B, _p(gx), _p(gc), _p(w_hh), _p(h0), _p(c0), _p(lens_dev), _p(hs_all[1:]),
                                         _p(gx) if save else None, _p(cs), _p(hN), _p(cN), _p(xch), _p(sync)))
        outs.append((hs_all, cs, hN, cN))
        flops += 8.0 * T * B * H * H
    with _Timed('lstm_seq_fwd', flops):
        check(lib.a4nt_lstm_seq_fwd_pair(H, ctypes.byref(probs[0]), ctypes.byref(probs[1]), _s()), 'a4nt_lstm_seq_fwd_pair')
    return outs


def lstm_seq_bwd_pair(reqs):
    """Two backward sweeps in one launch.  reqs: two tuples (w_hh, gates, cs, c0, lens_dev, dhs, dhN, dcN, need_h0)
    as for lstm_seq_bwd.  Returns two (dgates, dh0, dc0).  Under a backward budget (BWD_MAX_WGS) that does not hold
    the pair the two sweeps run one after the other."""
    lib = _lib.load()
    H = reqs[0][1].shape[2] // 4
    dev = reqs[0][1].device
    with _bwd_budget():
        geo = lstm_pair_geometry(reqs[0][1].shape[1], reqs[1][1].shape[1], H)
        if geo is not None:
            mt = geo[0]
            probs, outs = [], []
            flops = 0.0
            for slot, (w_hh, gates, cs, c0, lens_dev, dhs, dhN, dcN, need_h0) in enumerate(reqs):
                T, B, _ = gates.shape
                xch, sync, _g = _lstm_pair_ws(dev, B, H, mt, True, slot)
                dgates = torch.empty_like(gates)
                dh0 = torch.empty((B, H), device=dev, dtype=torch.float32) if need_h0 else None
                dc0 = torch.empty((B, H), device=dev, dtype=torch.float32) if need_h0 else None
                dbp = torch.empty((_g, 4 * H), device=dev, dtype=torch.float32) if H >= 256 else None
                probs.append(_lib.LstmBwdProblem(T, B, _p(w_hh), _p(gates), _p(cs), _p(c0), _p(lens_dev), _p(dhs), _p(dhN),
                                                 _p(dcN), _p(dgates), _p(dh0), _p(dc0), _p(xch), _p(sync), _p(dbp)))
                dgates._a4nt_dbias = dbp
                outs.append((dgates, dh0, dc0))
                flops += 8.0 * T * B * H * H
            with _Timed('lstm_seq_bwd', flops):
                check(lib.a4nt_lstm_seq_bwd_pair(H, ctypes.byref(probs[0]), ctypes.byref(probs[1]), _s()), 'a4nt_lstm_seq_bwd_pair')
            return outs
    return [lstm_seq_bwd(*r) for r in reqs]


# ----------------------------------------------------------------------------------------------
# autograd operators
# ----------------------------------------------------------------------------------------------
class _LinearKN(Function):
    """y = x W + b with W stored [K,N] (decoder_W layout, reference char_translator.py:93-95,279)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_b = b is not None
        return gemm_auto(x, W, bias=b)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        dx = gemm_auto(dy, W, transB=True) if ctx.needs_input_grad[0] else None
        if ctx.needs_input_grad[1] and ctx.has_b and ctx.needs_input_grad[2]:
            dW, db = gemm_tn_colsum(x, dy)
        else:
            dW = gemm_auto(x, dy, transA=True) if ctx.needs_input_grad[1] else None
            db = colsum(dy) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        return dx, dW, db


def linear_kn(x, W, b=None):
    return _LinearKN.apply(x, W, b)


class _LinearNK(Function):
    """y = x W^T + b with W stored [N,K] (nn.LSTM weight_ih layout)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_b = b is not None
        return gemm_auto(x, W, transB=True, bias=b)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        dx = gemm_auto(dy, W) if ctx.needs_input_grad[0] else None
        dW = gemm_auto(dy, x, transA=True) if ctx.needs_input_grad[1] else None
        db = colsum(dy) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        return dx, dW, db


def linear_nk(x, W, b=None):
    return _LinearNK.apply(x, W, b)


class _LstmLayer(Function):
    """One LSTM layer over a padded, time-major batch with per-row lengths
    (nn.LSTM over a PackedSequence in the reference: char_translator.py:175-198, char_lstm.py:142-164)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h0, c0, gc, lens_dev):
        T, B, I = x.shape
        H = w_hh.shape[1]
        x2 = x.reshape(T * B, I)
        gx = gemm(x2, w_ih, transB=True, bias=(b_ih + b_hh)).view(T, B, 4 * H)
        hs_all, cs, hN, cN = lstm_seq_fwd(gx, w_hh.contiguous(), lens_dev, h0, c0, gc, save=True)
        ctx.save_for_backward(x2, w_ih, w_hh, gx, cs, hs_all, c0 if c0 is not None else torch.empty(0), lens_dev
                              if lens_dev is not None else torch.empty(0))
        ctx.dims = (T, B, I, H)
        ctx.flags = (c0 is not None, lens_dev is not None, gc is not None, h0 is not None)
        return hs_all[1:], hN, cN

    @staticmethod
    def backward(ctx, dout, dhN, dcN):
        x2, w_ih, w_hh, gates, cs, hs_all, c0, lens_dev = ctx.saved_tensors
        T, B, I, H = ctx.dims
        has_c0, has_lens, has_gc, has_h0 = ctx.flags
        need_h0 = has_h0 and (ctx.needs_input_grad[5] or ctx.needs_input_grad[6])
        dout = dout.contiguous() if dout is not None else None
        dhN = dhN.contiguous() if dhN is not None else None
        dcN = dcN.contiguous() if dcN is not None else None
        dgates, dh0, dc0 = lstm_seq_bwd(w_hh.contiguous(), gates, cs, c0 if has_c0 else None,
                                        lens_dev if has_lens else None, dout, dhN, dcN, need_h0)
        dG2 = dgates.view(T * B, 4 * H)
        dx = gemm_auto(dG2, w_ih).view(T, B, I) if ctx.needs_input_grad[0] else None
        dw_ih = gemm_auto(dG2, x2, transA=True) if ctx.needs_input_grad[1] else None
        dw_hh = gemm_auto(dG2, hs_all[:T].view(T * B, H), transA=True) if ctx.needs_input_grad[2] else None
        db = lstm_bias_grad(dgates) if (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]) else None
        dgc = colsum(dgates.view(T, B * 4 * H)).view(B, 4 * H) if (has_gc and ctx.needs_input_grad[7]) else None
        return dx, dw_ih, dw_hh, db, db, (dh0 if has_h0 else None), (dc0 if has_c0 else None), dgc, None


HIDDEN_SIZES = (64, 128, 256, 512, 1024)      # instantiations of the persistent sweeps (lstm.hip)


def h2d(t, device, dtype=None):
    """Host tensor (or list / scalar) -> device without stopping the host: torch's plain .to(device) from pageable
    memory ends in a stream synchronisation, which drains the GPU queue the host has built up; a pinned staging copy
    + non_blocking is an asynchronous DMA (the pinned block is kept by torch's host allocator until the copy's event
    has passed).  Tensors already on `device` pass through."""
    if not torch.is_tensor(t):
        t = torch.as_tensor(t, dtype=dtype)
    elif dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    device = torch.device(device)
    if t.device == device or device.type != 'cuda':
        return t.to(device)
    if t.device.type == 'cpu' and not t.is_pinned():
        t = t.pin_memory()
    return t.to(device, non_blocking=True)


def attach_host_lens(lens_dev, lens_host):
    """Remember the host copy of a device length vector that has already been read back (one read-back per
    generation, gen_decode._GenDecode.forward); host_lens() hands it to later consumers."""
    lens_dev._a4nt_host = lens_host
    return lens_dev


def host_lens(lens_dev):
    """Host copy of a length vector: the attached one when the producer read it back already, else a D2H copy."""
    h = getattr(lens_dev, '_a4nt_host', None)
    return h if h is not None else lens_dev.cpu()


def padded_hidden(H):
    """The kernel hidden size that serves a model hidden size H: the next instantiated size."""
    for hp in HIDDEN_SIZES:
        if H <= hp:
            return hp
    raise A4NTError('hidden size %d: the gfx950 persistent LSTM kernels go up to %d hidden units' % (H, HIDDEN_SIZES[-1]))


def pad_gate_rows(t, H, Hp):
    """[4H, ...] (gate blocks i,f,g,o of H rows) -> [4Hp, ...]: every gate block zero-padded to Hp rows."""
    if t is None or Hp == H:
        return t
    tail = t.shape[1:]
    t4 = t.reshape((4, H) + tuple(tail))
    pad = [0, 0] * len(tail) + [0, Hp - H]
    return torch.nn.functional.pad(t4, pad).reshape((4 * Hp,) + tuple(tail))


def pad_gate_cols(t, H, Hp):
    """[..., 4H] -> [..., 4Hp], same per-gate zero padding along the last dimension."""
    if t is None or Hp == H:
        return t
    lead = t.shape[:-1]
    return torch.nn.functional.pad(t.reshape(tuple(lead) + (4, H)), [0, Hp - H]).reshape(tuple(lead) + (4 * Hp,))


def pad_last(t, H, Hp):
    return t if (t is None or Hp == H) else torch.nn.functional.pad(t, [0, Hp - H])


def pad_lstm_params(w_ih, w_hh, b_ih, b_hh, H, Hp):
    """nn.LSTM parameters of hidden size H as parameters of hidden size Hp whose extra units have zero weights and
    biases: their pre-activations are 0, so c = 0.5*c + 0.5*tanh(0) stays 0 from c0 = 0 and h = 0.5*tanh(0) = 0
    exactly, and they feed nothing back (zero columns of W_hh).  Differentiable (padding's backward is a slice)."""
    return (pad_gate_rows(w_ih, H, Hp), pad_gate_rows(pad_last(w_hh, H, Hp), H, Hp), pad_gate_rows(b_ih, H, Hp),
            pad_gate_rows(b_hh, H, Hp))


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, lens_dev=None, h0=None, c0=None, gc=None):
    """Returns (out [T,B,H] zero-padded past each row's length, hN [B,H], cN [B,H]).
    A batch larger than one persistent sweep holds (512 rows at H = 512, 256 at H = 1024) runs as several
    sweeps over row chunks: rows are independent.  A hidden size the kernels are not instantiated for (nn.LSTM in
    the reference takes any, char_translator.py:78, char_lstm.py:59) runs at the next instantiated size with
    zero-weight extra units, which stay exactly zero; the results are sliced back."""
    x = _f32(x, 'lstm input')
    B, H = x.shape[1], w_hh.shape[1]
    Hp = padded_hidden(H)
    if Hp != H:
        pw = pad_lstm_params(w_ih, w_hh, b_ih, b_hh, H, Hp)
        out, hN, cN = lstm_layer(x, pw[0], pw[1], pw[2], pw[3], lens_dev, pad_last(h0, H, Hp), pad_last(c0, H, Hp),
                                 pad_gate_cols(gc, H, Hp))
        return out[..., :H], hN[:, :H], cN[:, :H]
    sched = getattr(_pair_local, 'sched', None)
    if sched is not None:
        res = sched.request(_pair_local.me, (x.contiguous(), w_ih, w_hh, b_ih, b_hh, h0, c0, gc, lens_dev))
        if res is not None:
            return res
    if lstm_rows_fit(B, H):
        return _LstmLayer.apply(x.contiguous(), w_ih, w_hh, b_ih, b_hh, h0, c0, gc, lens_dev)
    step = B
    while step > 1 and not lstm_rows_fit(step, H):
        step = (step + 1) // 2
    step = max(16, step // 16 * 16)
    sl = lambda t_, a, b: None if t_ is None else t_[a:b]
    parts = [_LstmLayer.apply(x[:, a:a + step].contiguous(), w_ih, w_hh, b_ih, b_hh, sl(h0, a, a + step),
                              sl(c0, a, a + step), sl(gc, a, a + step), sl(lens_dev, a, a + step))
             for a in range(0, B, step)]
    return (torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 0), torch.cat([p[2] for p in parts], 0))


class _LstmLayerPair(Function):
    """Two independent _LstmLayer problems whose sweeps (forward, and later backward) share one launch.
    Inputs: the nine _LstmLayer inputs of problem 0, then those of problem 1; outputs likewise."""

    @staticmethod
    def forward(ctx, *args):
        reqs, saved, meta = [], [], []
        for k in range(2):
            x, w_ih, w_hh, b_ih, b_hh, h0, c0, gc, lens_dev = args[9 * k:9 * k + 9]
            T, B, I = x.shape
            H = w_hh.shape[1]
            x2 = x.reshape(T * B, I)
            gx = gemm(x2, w_ih, transB=True, bias=(b_ih + b_hh)).view(T, B, 4 * H)
            reqs.append((gx, w_hh.contiguous(), lens_dev, h0, c0, gc))
            meta.append(((T, B, I, H), (c0 is not None, lens_dev is not None, gc is not None, h0 is not None)))
            saved.append([x2, w_ih, w_hh, gx, None, None, c0 if c0 is not None else torch.empty(0),
                          lens_dev if lens_dev is not None else torch.empty(0)])
        res = lstm_seq_fwd_pair(reqs, save=True)
        outs = []
        for k in range(2):
            hs_all, cs, hN, cN = res[k]
            saved[k][4], saved[k][5] = cs, hs_all
            outs += [hs_all[1:], hN, cN]
        ctx.save_for_backward(*(saved[0] + saved[1]))
        ctx.meta = meta
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        sv = ctx.saved_tensors
        reqs, info = [], []
        for k in range(2):
            x2, w_ih, w_hh, gates, cs, hs_all, c0, lens_dev = sv[8 * k:8 * k + 8]
            (T, B, I, H), (has_c0, has_lens, has_gc, has_h0) = ctx.meta[k]
            nig = ctx.needs_input_grad[9 * k:9 * k + 9]
            need_h0 = has_h0 and (nig[5] or nig[6])
            dout, dhN, dcN = [d.contiguous() if d is not None else None for d in douts[3 * k:3 * k + 3]]
            reqs.append((w_hh.contiguous(), gates, cs, c0 if has_c0 else None, lens_dev if has_lens else None, dout,
                         dhN, dcN, need_h0))
            info.append((x2, w_ih, hs_all, T, B, I, H, has_c0, has_gc, has_h0, nig))
        need = [any(ctx.needs_input_grad[9 * k:9 * k + 9]) for k in range(2)]
        if need[0] and need[1]:
            res = lstm_seq_bwd_pair(reqs)
        else:                       # one side was run without a graph (mixed pair): only the other one has a backward sweep
            res = [lstm_seq_bwd(*reqs[k]) if need[k] else None for k in range(2)]
        grads = []
        for k in range(2):
            if not need[k]:
                grads += [None] * 9
                continue
            dgates, dh0, dc0 = res[k]
            x2, w_ih, hs_all, T, B, I, H, has_c0, has_gc, has_h0, nig = info[k]
            dG2 = dgates.view(T * B, 4 * H)
            dx = gemm_auto(dG2, w_ih).view(T, B, I) if nig[0] else None
            dw_ih = gemm_auto(dG2, x2, transA=True) if nig[1] else None
            dw_hh = gemm_auto(dG2, hs_all[:T].view(T * B, H), transA=True) if nig[2] else None
            db = lstm_bias_grad(dgates) if (nig[3] or nig[4]) else None
            dgc = colsum(dgates.view(T, B * 4 * H)).view(B, 4 * H) if (has_gc and nig[7]) else None
            grads += [dx, dw_ih, dw_hh, db, db, (dh0 if has_h0 else None), (dc0 if has_c0 else None), dgc, None]
        return tuple(grads)


# Pairing of the LSTM layers of two independent computations (adv_step.g_update: the discriminator over the
# generated batch and the reconstruction pass over the same batch).  The two computations run as coroutines
# (threads that hand one baton back and forth, never concurrently): each runs until its next lstm_layer call;
# when both have one pending and the pair fits one launch, _LstmLayerPair runs both.
class _PairScheduler:
    def __init__(self):
        self.cv = threading.Condition()
        self.turn = 0
        self.done = [False, False]
        self.pending = [None, None]
        self.grad = [True, True]          # grad mode of each side at its pending request
        self.result = [None, None]
        self.error = [None, None]

    def _other_waiting(self, me):
        return self.pending[1 - me] is not None and not self.done[1 - me]

    def request(self, me, args):
        """Called (holding the baton) from lstm_layer inside computation `me`."""
        other = 1 - me
        with self.cv:
            if self.done[other]:
                return None                                   # alone: the caller runs the single-problem layer
            if self._other_waiting(me):
                a = (args, self.pending[other]) if me == 0 else (self.pending[other], args)
                if _pairable(a[0], a[1]):
                    gm = [False, False]
                    gm[me], gm[other] = torch.is_grad_enabled(), self.grad[other]
                    if gm[0] == gm[1]:
                        outs = _LstmLayerPair.apply(*(a[0] + a[1]))
                    else:
                        # one side runs without a graph: its inputs enter detached (no gradients, no backward sweep for
                        # it) and its outputs leave detached; the other side's graph is recorded as usual
                        det = lambda tup: tuple(t.detach() if torch.is_tensor(t) else t for t in tup)
                        a = tuple(a[k] if gm[k] else det(a[k]) for k in range(2))
                        with torch.enable_grad():
                            outs = _LstmLayerPair.apply(*(a[0] + a[1]))
                        outs = tuple(o if gm[i // 3] else o.detach() for i, o in enumerate(outs))
                    mine, theirs = (outs[:3], outs[3:]) if me == 0 else (outs[3:], outs[:3])
                    self.pending[other] = None
                    self.result[other] = theirs
                    return mine                               # keep the baton: run on to the next request
                # not pairable: the other side's layer runs alone (by the other side), then it comes back here
                self.pending[other] = None
                self.result[other] = 'solo'
            # hand the baton over with my request pending
            self.pending[me] = args
            self.grad[me] = torch.is_grad_enabled()
            self.turn = other
            self.cv.notify_all()
            while self.turn != me:
                self.cv.wait()
            res, self.result[me] = self.result[me], None
            if res is None or res == 'solo':                  # the other side finished, or declined the pair
                self.pending[me] = None
                return None
            return res

    def finish(self, me, err=None):
        with self.cv:
            self.done[me] = True
            self.error[me] = err
            self.turn = 1 - me
            self.cv.notify_all()


def _pairable(a, b):
    xa, xb = a[0], b[0]
    Ha, Hb = a[2].shape[1], b[2].shape[1]
    if Ha != Hb or xa.device != xb.device:
        return False
    return lstm_pair_geometry(xa.shape[1], xb.shape[1], Ha) is not None


_pair_local = threading.local()


def paired(fn0, fn1):
    """Run fn0() and fn1() -- two computations none of which uses the other's results -- so that their LSTM
    layers are launched pairwise.  Returns (fn0(), fn1()).  With pairing switched off (PAIR_LSTM = False, or
    nested use) simply calls them in order."""
    if not PAIR_LSTM or getattr(_pair_local, 'sched', None) is not None:
        return fn0(), fn1()
    sched = _PairScheduler()
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None
    stream = torch.cuda.current_stream() if dev is not None else None     # torch's current stream is thread-local
    grad_mode = torch.is_grad_enabled()
    out = [None, None]

    def run(me, fn):
        with sched.cv:
            while sched.turn != me:
                sched.cv.wait()
        _pair_local.sched, _pair_local.me = sched, me
        err = None
        try:
            if dev is not None:
                torch.cuda.set_device(dev)
            with torch.set_grad_enabled(grad_mode):
                if stream is not None:
                    with torch.cuda.stream(stream):
                        out[me] = fn()
                else:
                    out[me] = fn()
        except BaseException as e:      # re-raised in the caller's thread
            err = e
        finally:
            _pair_local.sched = None
            sched.finish(me, err)

    threads = [threading.Thread(target=run, args=(k, f), daemon=True) for k, f in enumerate((fn0, fn1))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in sched.error:
        if e is not None:
            raise e
    return out[0], out[1]


PAIR_LSTM = True


def pairing_available():
    """True when paired() would interleave its two computations (not switched off, not already inside a pair)."""
    return bool(PAIR_LSTM) and getattr(_pair_local, 'sched', None) is None


class _Embed(Function):
    """nn.Embedding lookup with padding_idx semantics (zero gradient for the padding row)."""

    @staticmethod
    def forward(ctx, ids, W, padding_idx):
        ctx.save_for_backward(ids)
        ctx.shape = tuple(W.shape)
        ctx.padding_idx = padding_idx
        return embed_fwd(ids, W)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        dW = torch.zeros(ctx.shape, device=dout.device, dtype=torch.float32)
        embed_bwd(ids, dout, dW, None, ctx.padding_idx)
        return None, dW, None


def embedding(ids, W, padding_idx=-1):
    return _Embed.apply(ids, W, padding_idx)


class _ProjNLL(Function):
    """mean over valid rows of NLL(log_softmax(h W + b), target): fused vocab projection +
    log-softmax/NLL (+ the CE backward seed, left in place of the logits)."""

    @staticmethod
    def forward(ctx, h, W, b, target, n_valid):
        lib = _lib.load()
        N, V = h.shape[0], W.shape[1]
        K = h.shape[1]
        logits = padded_rows(N, V, h.device)          # 16-byte aligned rows: vector loads in the backward GEMMs
        nparts = int(lib.a4nt_sgemm_nll_parts(V))
        part = torch.empty((N, nparts, 2), device=h.device, dtype=torch.float32)
        W2 = _rowmajor(_f32(W, 'proj_nll W'))
        # projection whose epilogue leaves the per-row softmax statistics: the logits are written once ...
        with _Timed('sgemm_nn_128' if TIMER is not None else None, 2.0 * N * V * K):
            check(lib.a4nt_sgemm_nll(N, V, K, _p(h), _ld(h), _p(W2), _ld(W2), _p(logits), _ld(logits), _p(b), _p(part), _s()),
                  'a4nt_sgemm_nll')
        loss_rows = torch.empty(N, device=h.device, dtype=torch.float32)
        # ... and read once: loss from the merged statistics, (softmax - onehot)/n_valid in place
        check(lib.a4nt_nll_grad_rows(N, V, _p(logits), _ld(logits), _p(target), _p(part), nparts, 1.0 / float(n_valid),
                                     _p(loss_rows), 1, _s()), 'a4nt_nll_grad_rows')
        ctx.save_for_backward(h, W, logits)
        return loss_rows.sum() / float(n_valid)

    @staticmethod
    def backward(ctx, g):
        h, W, dlogits = ctx.saved_tensors
        # [N,V] x [V,H]: few output tiles over a long K -- the split-K model picks 3 slabs at N=12800 (1.35 -> 1.08 ms)
        dh = gemm_auto(dlogits, W, transB=True) if ctx.needs_input_grad[0] else None
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            dW, db = gemm_tn_colsum(h, dlogits)        # d(logits) is staged once for both
        else:
            dW = gemm_auto(h, dlogits, transA=True) if ctx.needs_input_grad[1] else None
            db = colsum(dlogits) if ctx.needs_input_grad[2] else None
        return (dh * g if dh is not None else None, dW * g if dW is not None else None,
                db * g if db is not None else None, None, None)


def proj_nll(h, W, b, target, n_valid):
    """h [N,H]; target [N] int64 with -1 for rows to ignore; n_valid = number of counted rows."""
    return _ProjNLL.apply(h.contiguous(), W, b, target.contiguous(), n_valid)


def gumbel_rows(logits, U, seed, step, scale, tau, hard, greedy, y, stats, tok, val_in, val_out, gen_lens, done,
                end_c, dense=None):
    lib = _lib.load()
    B, V = logits.shape
    check(lib.a4nt_gumbel_fwd(B, V, _p(logits), _ld(logits), _p(U), V if U is not None else 0, int(seed), int(step),
                              float(scale), float(tau), int(hard), int(greedy), _p(y), V, _p(stats), _p(tok),
                              _p(val_in), _p(val_out), _p(gen_lens), _p(done), int(end_c), _p(dense),
                              V if dense is not None else 0, _s()), 'a4nt_gumbel_fwd')


def gumbel_bwd_rows(ds, y, stats, coef, inv_tau=1.0, out=None):
    """out (default: ds, in place) = coef * p * (ds - <p, ds>) per row."""
    lib = _lib.load()
    B, V = ds.shape
    if out is None:
        out = ds
    check(lib.a4nt_gumbel_bwd(B, V, _p(ds), _ld(ds), _p(y), _ld(y), _p(stats), float(coef), float(inv_tau), _p(out),
                              _ld(out), _s()), 'a4nt_gumbel_bwd')


def proj_gumbel_pack(W):
    """Packed image of the [K,N] projection matrix for proj_gumbel(packed=...): one pass over W, made once per
    generation."""
    lib = _lib.load()
    W = _rowmajor(_f32(W, 'proj_gumbel_pack W'))
    K, N = W.shape
    Wp = torch.empty(int(lib.a4nt_proj_gumbel_packed_floats(K, N)), device=W.device, dtype=torch.float32)
    check(lib.a4nt_proj_gumbel_pack(K, N, _p(W), _ld(W), _p(Wp), _s()), 'a4nt_proj_gumbel_pack')
    return Wp


def proj_gumbel(h, W, bias, U, seed, step, scale, tau, hard, greedy, z, part, stats, tok, val_in, val_out, gen_lens,
                done, end_c, zrow0=0, packed=None):
    """Fused decoder-step projection + Gumbel-softmax sampling (a4nt_proj_gumbel_fwd).
    z (optional) receives rows [zrow0, zrow0 + z.shape[0]) of the batch.  packed: W's image from proj_gumbel_pack
    (W is then only used for its shape)."""
    lib = _lib.load()
    M, K = h.shape
    N = W.shape[1]
    with _Timed('proj_gumbel', 2.0 * M * N * K):
        check(lib.a4nt_proj_gumbel_fwd(M, N, K, _p(h), _ld(h), _p(W if packed is None else packed), _ld(W) if packed is None else 0, _p(bias), float(scale), float(tau),
                                       int(hard), int(greedy), _p(U), _ld(U) if U is not None else 0, int(seed),
                                       int(step), _p(z), _ld(z) if z is not None else 0, int(zrow0),
                                       int(z.shape[0]) if z is not None else 0, _p(part), _p(stats), _p(tok),
                                       _p(val_in), _p(val_out), _p(gen_lens), _p(done), int(end_c), _s()),
              'a4nt_proj_gumbel_fwd')


class _GumbelDense(Function):
    """gumbel_softmax_sample on a [B,V] logit matrix returning the dense sample (generic path:
    soft (non straight-through) samples, stacked decoders).  x is already scaled by softmax_scale."""

    @staticmethod
    def forward(ctx, x, U, seed, step, tau, hard):
        B, V = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        stats = torch.empty((B, 2), device=x.device, dtype=torch.float32)
        tok = torch.empty(B, device=x.device, dtype=torch.int64)
        dense = torch.empty_like(x)
        gumbel_rows(x, U, seed, step, 1.0, tau, hard, 0, y, stats, tok, None, None, None, None, -1, dense)
        ctx.save_for_backward(y, stats)
        ctx.tau = tau
        ctx.mark_non_differentiable(tok)
        return dense, tok

    @staticmethod
    def backward(ctx, dd, _dtok):
        y, stats = ctx.saved_tensors
        ds = dd.contiguous().clone()
        gumbel_bwd_rows(ds, y, stats, 1.0 / ctx.tau)
        return ds, None, None, None, None, None


def gumbel_dense(x, U, seed, step, tau, hard):
    return _GumbelDense.apply(x, U, seed, step, tau, hard)


# ----------------------------------------------------------------------------------------------
# optimizer step as multi-tensor launches (csrc/optim.hip)
# ----------------------------------------------------------------------------------------------
_mt_plans = {}


def _mt_plan(dev, numels):
    """Chunk map of a tensor list (depends on the sizes only: built and uploaded once)."""
    key = (dev.index, tuple(numels))
    if key not in _mt_plans:
        ch = int(_lib.load().a4nt_mt_chunk_elems())
        rows = [(ti, c) for ti, n in enumerate(numels) for c in range((n + ch - 1) // ch)]
        m = torch.tensor(rows, dtype=torch.int32).reshape(-1, 2)
        _mt_plans[key] = (h2d(m, dev).contiguous(), len(rows), torch.empty(max(1, len(rows)), device=dev))
    return _mt_plans[key]


def _mt_ok(ts):
    return all(t is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and not t.is_sparse for t in ts)


def _mt_table(rows, dev):
    """[p, g, sq, n] rows -> device (pinned staging: the host does not wait for the GPU)."""
    return h2d(torch.tensor(rows, dtype=torch.int64), dev)


def clip_grad_norm_(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_ (2-norm) in three launches: per-chunk sums of squares, norm + clipping factor on
    the device, in-place scaling when the factor is below one.  Returns the total norm (0-dim device tensor)."""
    ps = [p for p in parameters if p.grad is not None]
    grads = [p.grad for p in ps]
    if not grads or not _mt_ok(grads) or len({g.device for g in grads}) != 1:
        return torch.nn.utils.clip_grad_norm_(ps, max_norm)
    lib = _lib.load()
    dev = grads[0].device
    cmap, nchunks, partial = _mt_plan(dev, [g.numel() for g in grads])
    table = _mt_table([[0, g.data_ptr(), 0, g.numel()] for g in grads], dev)
    out2 = torch.empty(2, device=dev)
    s = _s()
    check(lib.a4nt_mt_sumsq(nchunks, _p(table), _p(cmap), _p(partial), s), 'a4nt_mt_sumsq')
    check(lib.a4nt_mt_clip_coef(nchunks, _p(partial), float(max_norm), _p(out2), s), 'a4nt_mt_clip_coef')
    check(lib.a4nt_mt_scale(nchunks, _p(table), _p(cmap), _p(out2), s), 'a4nt_mt_scale')
    return out2[0]


class RMSprop(torch.optim.RMSprop):
    """torch.optim.RMSprop (same constructor, param_groups and state_dict: 'step', 'square_avg') whose step is ONE
    launch per parameter group on GPU tensors (a4nt_mt_rmsprop).  Options the kernel does not cover (momentum, centered,
    maximize, non-fp32 / non-contiguous / CPU tensors) take torch's implementation."""

    @torch.no_grad()
    def step(self, closure=None):
        plain = all(g['momentum'] == 0 and not g['centered'] and not g.get('maximize', False) and
                    not g.get('capturable', False) and not g.get('differentiable', False) for g in self.param_groups)
        todo = []
        for g in self.param_groups:
            ps = [p for p in g['params'] if p.grad is not None]
            if ps:
                todo.append((g, ps))
        if closure is not None or not plain or not all(_mt_ok(ps) and _mt_ok([p.grad for p in ps]) for _, ps in todo):
            return super().step(closure)
        lib = _lib.load()
        for g, ps in todo:
            rows = []
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.zeros((), dtype=torch.float32)
                    st['square_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                if not _mt_ok([st['square_avg']]):
                    raise A4NTError('RMSprop: square_avg must be a contiguous fp32 GPU tensor')
                rows.append([p.data_ptr(), p.grad.data_ptr(), st['square_avg'].data_ptr(), p.numel()])
            dev = ps[0].device
            cmap, nchunks, _ = _mt_plan(dev, [p.numel() for p in ps])
            table = _mt_table(rows, dev)
            check(lib.a4nt_mt_rmsprop(nchunks, _p(table), _p(cmap), float(g['lr']), float(g['alpha']), float(g['eps']),
                                      float(g['weight_decay']), _s()), 'a4nt_mt_rmsprop')
        return None


class _BceLogitsCol(Function):
    """mean_n w_n * BCEWithLogits(logits[n, col], target): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, logits, col, target, weight):
        lib = _lib.load()
        x = _f32(logits, 'bce logits').contiguous()
        w = None if weight is None else _f32(weight, 'bce weight').contiguous()
        N, C = x.shape
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        check(lib.a4nt_bce_logits_fwd(N, C, int(col), _p(x), float(target), _p(w), _p(loss), _s()), 'a4nt_bce_logits_fwd')
        ctx.save_for_backward(x, w)
        ctx.col, ctx.target = int(col), float(target)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        N, C = x.shape
        dx = torch.empty_like(x)
        g = g.contiguous().view(1)
        check(_lib.load().a4nt_bce_logits_bwd(N, C, ctx.col, _p(x), ctx.target, _p(w), _p(g), _p(dx), _s()),
              'a4nt_bce_logits_bwd')
        return dx, None, None, None


def bce_with_logits_col(logits, col, target, weight=None):
    """F.binary_cross_entropy_with_logits(logits[:, col], full_like(.., target), weight=weight) for [N, C] logits on the
    GPU (anything else: torch's own)."""
    if logits.is_cuda and logits.dim() == 2 and logits.dtype == torch.float32 and (weight is None or weight.is_cuda):
        return _BceLogitsCol.apply(logits, col, target, weight)
    x = logits[:, col]
    return torch.nn.functional.binary_cross_entropy_with_logits(x, torch.full_like(x, float(target)), weight=weight)
