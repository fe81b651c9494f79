#!/usr/bin/env python3
"""Every time-batched GEMM of one adversarial training step with its shape, in call order (python-side calls of
a4nt_sgemm / a4nt_sgemm_nll; the two decoding loops' per-step products are issued inside the library and not listed)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import a4nt_amd as impl
from a4nt_amd import _lib

lib = _lib.load()
log = []
orig, orig_nll, orig_cs = lib.a4nt_sgemm, lib.a4nt_sgemm_nll, lib.a4nt_sgemm_tn_colsum


def wrap(tA, tB, M, N, K, *rest):
    log.append(('sgemm', tA, tB, M, N, K, rest[8]))
    return orig(tA, tB, M, N, K, *rest)


def wrap_nll(M, N, K, *rest):
    log.append(('sgemm_nll', 0, 0, M, N, K, 1))
    return orig_nll(M, N, K, *rest)


def wrap_cs(M, N, K, *rest):
    log.append(('tn_colsum', 1, 0, M, N, K, rest[6]))
    return orig_cs(M, N, K, *rest)


lib.a4nt_sgemm, lib.a4nt_sgemm_nll, lib.a4nt_sgemm_tn_colsum = wrap, wrap_nll, wrap_cs
torch.manual_seed(0)
batch = 128
G = impl.CharTranslator(bench.gen_params(batch)).cuda().train()
D = impl.CharLstm(bench.eval_params()).cuda().train()
optG, optD = impl.make_optimizers(G, D)
batches = bench.make_batches(impl, 1234, batch, 6)
for it in range(2):
    del log[:]
    bench.adv_step(impl, G, D, optG, optD, batches, it, batch, (None, None))
torch.cuda.synchronize()
tot = 0.0
for i, (name, tA, tB, M, N, K, ks) in enumerate(log):
    gf = 2.0 * M * N * K / 1e9
    tot += gf
    if gf >= 1.0:
        print('%3d %-9s %s%s  M=%6d N=%6d K=%6d ks=%2d  %7.1f GFLOP' % (i, name, 'T' if tA else 'N', 'T' if tB else 'N', M, N, K, ks, gf))
print('%d calls, %.1f GFLOP in the python-side products' % (len(log), tot))
