#!/usr/bin/env python3
"""Host time per launch of the wrappers used inside the per-step decode loops (GPU box only): tiny problems, so that
the GPU is never the limit; the loop is timed on the host without synchronising until the end."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd
from a4nt_amd import ops, _lib
lib = _lib.load()
dev = 'cuda'
A = torch.randn(16, 64, device=dev); B = torch.randn(48, 64, device=dev); C = torch.zeros(16, 48, device=dev)
slabs = torch.zeros(4, 16, 48, device=dev)
ds = torch.randn(16, 48, device=dev); y = torch.randn(16, 48, device=dev); st = torch.ones(16, 2, device=dev); out = torch.zeros(16, 48, device=dev)


def bench(name, fn, n=3000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-34s host %.1f us per call (GPU drained %.1f ms after the loop)' % (name, (t1 - t0) / n * 1e6, (t2 - t1) * 1e3))


bench('ops.gemm NT out=', lambda: ops.gemm(A, B, transB=True, out=C))
bench('ops.gemm NT out= beta=1 (views)', lambda: ops.gemm(A[:, 16:], B[:, 16:], transB=True, out=C[:8], beta=1))
bench('ops.gemm split-K slabs', lambda: ops.gemm(A, B, transB=True, out=slabs, ksplit=4))
bench('ops.gumbel_bwd_rows', lambda: ops.gumbel_bwd_rows(ds, y, st, 1.0, 2.0, out=out))
bench('torch add_ (reference point)', lambda: C.add_(1.0))
s = ops._s()
bench('raw ctypes a4nt_slab_reduce', lambda: lib.a4nt_slab_reduce(4, 16 * 48, 16, 48, ops._p(slabs), 48, 0, None, 0, ops._p(C), 48, s))
bench('ops._s()', lambda: ops._s(), n=20000)
bench('ops._p(tensor)', lambda: ops._p(C), n=20000)
