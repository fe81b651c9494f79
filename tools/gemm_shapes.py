#!/usr/bin/env python3
"""Per-shape time of every a4nt_sgemm launch in one adversarial step (bench.py's workload)."""
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import a4nt_amd as impl
from a4nt_amd import ops


def main():
    torch.manual_seed(0)
    G, D = impl.CharTranslator(bench.gen_params(bench.B)), impl.CharLstm(bench.eval_params())
    G.train(); D.train()
    optG, optD = impl.make_optimizers(G, D)
    batches = bench.make_batches(impl, 1234, bench.B, 6)
    for it in range(2):
        bench.adv_step(impl, G, D, optG, optD, batches, it, bench.B, (None, None))
    rec = []
    orig = ops.gemm

    def gemm(A, Bm, transA=False, transB=False, bias=None, out=None, beta=0, ksplit=1):
        M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
        N = Bm.shape[0] if transB else Bm.shape[1]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(A, Bm, transA, transB, bias=bias, out=out, beta=beta, ksplit=ksplit)
        e1.record()
        rec.append(((int(transA), int(transB), M, N, K, int(ksplit)), e0, e1))
        return r

    ops.gemm = gemm
    bench.adv_step(impl, G, D, optG, optD, batches, 2, bench.B, (None, None))
    torch.cuda.synchronize()
    ops.gemm = orig
    acc = defaultdict(lambda: [0, 0.0])
    for key, e0, e1 in rec:
        acc[key][0] += 1
        acc[key][1] += e0.elapsed_time(e1)
    tot = sum(v[1] for v in acc.values())
    print('total gemm ms/step %.2f' % tot)
    for key, (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        ta, tb, M, N, K, ks = key
        print('tA=%d tB=%d M=%6d N=%6d K=%6d ks=%2d  x%4d  %8.1f us each  %6.2f ms  %6.1f TF' %
              (ta, tb, M, N, K, ks, n, 1000 * ms / n, ms, 2.0 * M * N * K * n / ms / 1e9))


if __name__ == '__main__':
    main()
