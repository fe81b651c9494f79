#!/usr/bin/env python3
"""Experiment: does the latency-bound reverse loop of the free-running decoder (100 steps x 6 short kernels) run
beside independent weight-gradient GEMMs on a second stream?  Times the loop alone, the GEMMs alone, and both
together (loop on a high-priority stream)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd  # noqa: F401
from a4nt_amd import ops, _lib

lib = _lib.load()
dev = torch.device('cuda')
B, H, E, V, n = 128, 512, 512, 10001, 100
ops._ensure_init()
g = torch.Generator(device='cuda').manual_seed(1)
r = lambda *s, sc=0.1: torch.randn(*s, device=dev, generator=g) * sc
S2, D2 = ops.padded_rows(n * B, V, dev), ops.padded_rows(n * B, V, dev)
S2.normal_(generator=g)
ybuf = r(n, B, V)
stats = torch.ones(n, B, 2, device=dev)
stats[..., 1] = 100.0
emb_W, W_out, Wcat = r(V, E), r(H, V), r(4 * H, H + E)
gates, cs = torch.sigmoid(r(n, B, 4 * H, sc=1.0)), r(n, B, H)
dc, dG, rec = torch.zeros(B, H, device=dev), torch.empty(n, B, 4 * H, device=dev), torch.empty(n, B, H + E, device=dev)
ks_v = max(1, min(16, ops._auto_ksplit(B, H, V)))
ks_r = max(1, ops._auto_ksplit(B, H + E, 4 * H))
dh_slabs = torch.empty(ks_v, B, H, device=dev)
rec_slabs = torch.empty(ks_r, B, H + E, device=dev)
ba = _lib.GenBwdArgs()
p = ops._p
ba.n, ba.B, ba.H, ba.E, ba.V = n, B, H, E, V
ba.S, ba.lds, ba.D, ba.ldd = p(S2), ops._ld(S2), p(D2), ops._ld(D2)
ba.y, ba.ldy, ba.stats = p(ybuf), V, p(stats)
ba.coef, ba.inv_tau = 6.0, 2.0
ba.emb_W, ba.W_out, ba.ldw, ba.Wcat = p(emb_W), p(W_out), V, p(Wcat)
ba.gates, ba.cs, ba.dc, ba.dG, ba.rec = p(gates), p(cs), p(dc), p(dG), p(rec)
ba.dh_slabs, ba.ks_v, ba.rec_slabs, ba.ks_r = p(dh_slabs), ks_v, p(rec_slabs), ks_r

# independent work: the reconstruction pass's weight gradients (two [2048 x 512] x K=12800 pairs and the [512 x 10001] one)
X, dGr, Hs = r(n * B, 512), r(n * B, 2048), r(n * B, 512)
Dr = ops.padded_rows(n * B, V, dev)
Dr.normal_(generator=g)


def gemms():
    for _ in range(2):
        ops.gemm_auto(dGr, X, transA=True)
        ops.gemm_auto(dGr, Hs, transA=True)
    ops.gemm_auto(Hs, Dr, transA=True)


def loop(stream):
    _lib.check(lib.a4nt_gen_loop_bwd(ctypes.byref(ba), stream.cuda_stream), 'a4nt_gen_loop_bwd')


def timed(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


main = torch.cuda.current_stream()
hi = torch.cuda.Stream(priority=-1)
print('reverse loop alone : %.2f ms' % timed(lambda: loop(main)))
print('GEMMs alone        : %.2f ms' % timed(gemms))


def both():
    hi.wait_stream(main)
    loop(hi)
    gemms()
    main.wait_stream(hi)


print('both, two streams  : %.2f ms' % timed(both))


# ---- CU-masked streams (hipExtStreamCreateWithCUMask): the GEMMs confined to part of the chip -------------------------
# r03 on the box: loop 6.57 ms, GEMMs 2.00 ms, two plain streams 8.82 ms; GEMMs masked to CUs 128-255 (3.83 ms alone)
# beside the loop on all CUs 8.91 ms -- the loop's one-workgroup-per-CU launches wait for a GEMM tile to leave each of
# those CUs; loop on CUs 0-127 (11.64 ms alone: its kernels are sized for 256 CUs) beside the GEMMs on the other half
# 12.62 ms.  Nothing to gain from deferring weight gradients beside the loop.
hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, 'hipExtStreamCreateWithCUMask -> %d' % rc
    return torch.cuda.ExternalStream(s.value)


FULL = 0xffffffff
for name, loop_mask, gemm_mask in (('loop on all CUs, GEMMs on CUs 128-255', None, [0, 0, 0, 0, FULL, FULL, FULL, FULL]),
                                   ('loop on CUs 0-127, GEMMs on CUs 128-255', [FULL] * 4 + [0] * 4, [0] * 4 + [FULL] * 4)):
    ls = hi if loop_mask is None else masked_stream(loop_mask)
    gs = masked_stream(gemm_mask)

    def gemms_on():
        with torch.cuda.stream(gs):
            gemms()

    def both_masked():
        ls.wait_stream(main)
        gs.wait_stream(main)
        loop(ls)
        gemms_on()
        main.wait_stream(ls)
        main.wait_stream(gs)

    def loop_only():
        ls.wait_stream(main)
        loop(ls)
        main.wait_stream(ls)

    def gemms_only():
        gs.wait_stream(main)
        gemms_on()
        main.wait_stream(gs)

    print('%s: loop alone %.2f ms, GEMMs alone %.2f ms, together %.2f ms'
          % (name, timed(loop_only), timed(gemms_only), timed(both_masked)))
