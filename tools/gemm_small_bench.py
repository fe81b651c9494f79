#!/usr/bin/env python3
"""64x64-tile GEMMs of one decoder step (128/256 rows) across k-group settings (GPU box only)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd
from a4nt_amd import ops, _lib
lib = _lib.load()
SHAPES = [  # name, M, N, K, tA, tB, ksplit, beta
    ('dec ds += demb*E^T', 128, 10001, 512, 0, 1, 1, 1),
    ('dec dh splitk16', 128, 512, 10001, 0, 1, 16, 0),
    ('dec dh splitk8', 128, 512, 10001, 0, 1, 8, 0),
    ('dec dh splitk32', 128, 512, 10001, 0, 1, 32, 0),
    ('dec rec splitk8', 128, 1024, 2048, 0, 0, 8, 0),
    ('dec rec splitk4', 128, 1024, 2048, 0, 0, 4, 0),
    ('dec rec splitk2', 128, 1024, 2048, 0, 0, 2, 0),
    ('head 128x2048x512', 128, 2048, 512, 0, 1, 1, 0),
    ('cg 256x2048x512', 256, 2048, 512, 0, 1, 1, 0),
]
for name, M, N, K, tA, tB, ks, beta in SHAPES:
    A = torch.randn((K, M) if tA else (M, K), device='cuda')
    B = torch.randn((N, K) if tB else (K, N), device='cuda')
    out = torch.zeros((ks, M, N) if ks > 1 else (M, N), device='cuda')
    ref = (A.t() if tA else A).double() @ (B.t() if tB else B).double()
    line = '%-20s M=%4d N=%5d K=%5d ks=%2d ' % (name, M, N, K, ks)
    for kg in (1, 2, 4, 0):
        _lib.check(lib.a4nt_sgemm_set_kgroups(kg), 'kg')
        out.zero_()
        r = ops.gemm(A, B, bool(tA), bool(tB), out=out, ksplit=ks, beta=0)
        got = (out.sum(0) if ks > 1 else out).double()
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, (name, kg, err)
        for _ in range(3):
            ops.gemm(A, B, bool(tA), bool(tB), out=out, ksplit=ks, beta=beta)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            ops.gemm(A, B, bool(tA), bool(tB), out=out, ksplit=ks, beta=beta)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        line += ' kg%d %6.1f us' % (kg, us)
    print(line, flush=True)
_lib.check(lib.a4nt_sgemm_set_kgroups(0), 'kg')

for M, N, K in [(128, 10001, 512), (256, 10001, 512), (64, 10001, 512), (128, 10001, 256)]:
    A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); out = torch.zeros(M, N, device='cuda')
    for beta in (0, 1):
        line = 'nt_rows M=%d N=%d K=%d beta=%d ' % (M, N, K, beta)
        for route in (False, True):
            ops.NT_ROWS = route
            for _ in range(3):
                ops.gemm(A, B, transB=True, out=out, beta=beta)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.gemm(A, B, transB=True, out=out, beta=beta)
            e1.record()
            torch.cuda.synchronize()
            line += ' %s %6.1f us' % ('nt_rows' if route else 'sgemm  ', e0.elapsed_time(e1) / 50 * 1e3)
        print(line, flush=True)
