#!/usr/bin/env python3
"""Column sums (bias gradients) at the step's shapes: time and HBM rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd
from a4nt_amd import ops
for M, N in ((12800, 2048), (25600, 2048), (12800, 10001), (100, 128 * 2048)):
    A = ops.padded_rows(M, N, torch.device('cuda'))
    A.normal_()
    ref = A.double().sum(0).float()
    out = ops.colsum(A)
    err = float((out - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        ops.colsum(A)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.colsum(A)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print('colsum %6d x %6d: %6.1f us  %5.2f TB/s  rel err %.1e' % (M, N, us, M * N * 4 / us / 1e6, err))
