#!/usr/bin/env python3
"""Micro-benchmark of a4nt_sgemm on the GEMM shapes of the adversarial step (GPU box only)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd
from a4nt_amd import ops

SHAPES = [  # name, M, N, K, tA, tB, ksplit
    ('cube4096 NN', 4096, 4096, 4096, 0, 0, 1),
    ('cube4096 NT', 4096, 4096, 4096, 0, 1, 1),
    ('cube4096 TN', 4096, 4096, 4096, 1, 0, 1),
    ('gates x*Wih^T', 12800, 2048, 512, 0, 1, 1),
    ('vocab proj h*W', 12800, 10001, 512, 0, 0, 1),
    ('S_ext demb*E^T', 12800, 10001, 512, 0, 1, 1),
    ('dW_out h^T*dl', 512, 10001, 12800, 1, 0, 1),
    ('dh dl*W^T', 12800, 512, 10001, 0, 1, 1),
    ('dWih dG^T*x', 2048, 512, 12800, 1, 0, 4),
    ('dx dG*Wih', 12800, 512, 2048, 0, 0, 1),
    ('dec logits', 128, 10001, 512, 0, 0, 1),
    ('dec logits ks2', 128, 10001, 512, 0, 0, 2),
    ('dec logits ks4', 128, 10001, 512, 0, 0, 4),
    ('dec logits ks8', 128, 10001, 512, 0, 0, 8),
    ('dec ds', 128, 10001, 512, 0, 1, 1),
    ('dec dh splitk', 128, 512, 10001, 0, 1, 16),
    ('dec rec splitk', 128, 1024, 2048, 0, 0, 8),
    ('ptab E*Wie^T', 10001, 2048, 512, 0, 1, 1),
]
for name, M, N, K, tA, tB, ks in SHAPES:
    A = torch.randn((K, M) if tA else (M, K), device='cuda')
    B = torch.randn((N, K) if tB else (K, N), device='cuda')
    out = torch.empty((ks, M, N) if ks > 1 else (M, N), device='cuda')
    for _ in range(3):
        ops.gemm(A, B, bool(tA), bool(tB), out=out if ks == 1 else None, ksplit=ks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.gemm(A, B, bool(tA), bool(tB), out=out if ks == 1 else None, ksplit=ks)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print('%-18s M=%5d N=%5d K=%5d tA=%d tB=%d ks=%2d  %8.1f us  %6.1f TFLOP/s' % (name, M, N, K, tA, tB, ks, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
