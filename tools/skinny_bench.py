#!/usr/bin/env python3
"""a4nt_sgemm_skinny against a4nt_sgemm on the per-step shapes of the decoder (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd                                    # noqa: F401
from a4nt_amd import ops

SHAPES = [  # name, M, N, K, tB, ksplit of the LDS kernel
    ('ds   demb*E^T', 128, 10001, 512, 1, 1),
    ('dh   dl*W^T', 128, 512, 10001, 1, 32),
    ('rec  dG*Wcat', 128, 1024, 2048, 0, 8),
    ('logits h*W', 128, 10001, 512, 0, 1),
    ('cfg5 ds', 256, 30001, 512, 1, 1),
    ('cfg5 dh', 256, 1024, 30001, 1, 16),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, M, N, K, tB, ks0 in SHAPES:
    A = torch.randn(M, K, device='cuda')
    B = torch.randn((N, K) if tB else (K, N), device='cuda')
    t0 = timeit(lambda: ops.gemm(A, B, False, bool(tB), ksplit=ks0))
    res = []
    ksug = ops.skinny_ksplit(M, N, K)
    for ks in sorted(set([1, ksug, max(1, ksug // 2), ksug * 2])):
        try:
            res.append((ks, timeit(lambda: ops.gemm_skinny(A, B, bool(tB), ksplit=ks))))
        except Exception as e:
            res.append((ks, float('nan')))
    fl = 2.0 * M * N * K / 1e6
    print('%-14s M=%4d N=%6d K=%6d tB=%d | sgemm ks=%2d %6.1f us %5.1f TF | skinny (suggested ks=%d): %s' % (
        name, M, N, K, tB, ks0, t0, fl / t0, ksug, '  '.join('ks=%d %.1f us %.1f TF' % (k, t, fl / t) for k, t in res)))
