import os, sys, torch
sys.path.insert(0, '/root/repo')
import a4nt_amd
from a4nt_amd import ops
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, tA, tB) in ((2048, 512, 12800, 1, 0), (2048, 512, 25600, 1, 0), (12800, 512, 2048, 0, 0), (25600, 512, 2048, 0, 0), (12800, 512, 10001, 0, 1), (512, 10001, 12800, 1, 0)):
    A = torch.randn((K, M) if tA else (M, K), device='cuda')
    B = torch.randn((N, K) if tB else (K, N), device='cuda')
    line = 'M=%5d N=%5d K=%5d %s auto=%d:' % (M, N, K, 'TN' if tA else ('NT' if tB else 'NN'), ops._auto_ksplit(M, N, K))
    for ks in (1, 2, 3, 4, 6, 8, 12, 16):
        if ks == 1:
            us = t(lambda: ops.gemm(A, B, bool(tA), bool(tB)))
        else:
            us = t(lambda: ops.slab_reduce(ops.gemm(A, B, bool(tA), bool(tB), ksplit=ks)))
        line += '  ks%2d %6.1f' % (ks, us)
    print(line, flush=True)
