#!/usr/bin/env python3
"""The reverse loop's recurrent product dG_t [128 x 2048] * Wcat [2048 x 1024]: split-K slabs + reduction against
k-groups inside the workgroup (no slabs, no reduction launch), and the same for dlogits_t * W_out^T (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd  # noqa: F401
from a4nt_amd import ops, _lib

lib = _lib.load()


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K, tB, name) in ((128, 1024, 2048, 0, 'dG * Wcat'), (128, 512, 10001, 1, 'dlogits * W_out^T')):
    A = torch.randn(M, K, device='cuda')
    B = torch.randn((N, K) if tB else (K, N), device='cuda')
    print(name, M, N, K)
    for kg in (1, 2, 4):
        lib.a4nt_sgemm_set_kgroups(kg)
        line = '  k-groups %d:' % kg
        for ks in (1, 2, 4, 8, 16):
            if ks == 1:
                us = t(lambda: ops.gemm(A, B, False, bool(tB)))
            else:
                us = t(lambda: ops.slab_reduce(ops.gemm(A, B, False, bool(tB), ksplit=ks)))
            line += '  ks%2d %5.1f us' % (ks, us)
        print(line, flush=True)
    lib.a4nt_sgemm_set_kgroups(0)
