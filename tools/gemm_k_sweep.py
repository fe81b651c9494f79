#!/usr/bin/env python3
"""a4nt_sgemm (NT, 128x128 tiles) over tile counts and K depths: what a tile's fixed cost (first loads, epilogue) takes
from the K = 512 shapes of the step (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd  # noqa: F401
from a4nt_amd import ops


def rate(M, N, K, tA=0, tB=1, n=10):
    A = torch.randn((K, M) if tA else (M, K), device='cuda')
    B = torch.randn((N, K) if tB else (K, N), device='cuda')
    out = torch.empty((M, N), device='cuda')
    for _ in range(3):
        ops.gemm(A, B, bool(tA), bool(tB), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(A, B, bool(tA), bool(tB), out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9


print('NT, N = 2048: rows are M (tiles), columns K; us / TFLOP/s')
for M in (4096, 8192, 12288, 12800, 16384, 24576, 25600, 32768):
    line = 'M=%6d (%4d tiles)' % (M, (M // 128) * 16)
    for K in (256, 512, 1024, 2048, 4096):
        us, tf = rate(M, 2048, K)
        line += '  K=%4d %7.1f us %6.1f' % (K, us, tf)
    print(line)
print('NN, N = 512 (the data-gradient shape), K = 2048 and 512')
for M in (12800, 25600):
    for K in (512, 2048):
        us, tf = rate(M, 512, K, 0, 0)
        print('M=%6d K=%4d %7.1f us %6.1f TFLOP/s' % (M, K, us, tf))
