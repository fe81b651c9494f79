#!/usr/bin/env python3
"""Micro-benchmark of the fused projection + Gumbel kernel at the decoder-step shape (GPU box only).
usage: pg_bench.py [iters] [greedy 0/1] [write_z 0/1] [M] [packed 0/1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd                                    # noqa: F401
from a4nt_amd import ops, _lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
greedy = int(sys.argv[2]) if len(sys.argv) > 2 else 0
write_z = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = _lib.load()
M, N, K = int(sys.argv[4]) if len(sys.argv) > 4 else 128, 10001, 512
packed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
h = torch.randn(M, K, device='cuda')
W = torch.randn(K, N, device='cuda') * 0.05
b = torch.zeros(N, device='cuda')
z = ops.padded_rows(M, N, 'cuda') if write_z else None
part = torch.empty(int(lib.a4nt_proj_gumbel_part_floats(M, N)), device='cuda')
stats = torch.empty(M, 2, device='cuda')
tok = torch.zeros(M, dtype=torch.int64, device='cuda')
vi, vo = torch.ones(M, device='cuda'), torch.ones(M, device='cuda')
gl = torch.zeros(M, dtype=torch.int32, device='cuda')
done = torch.zeros(M, dtype=torch.uint8, device='cuda')
Wp = ops.proj_gumbel_pack(W) if packed else None
for it in range(5):
    ops.proj_gumbel(h, W, b, None, 1, it, 3.0, 0.5, True, bool(greedy), z, part, stats, tok, vi, vo, gl, done, -1, packed=Wp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(iters):
    ops.proj_gumbel(h, W, b, None, 1, it, 3.0, 0.5, True, bool(greedy), z, part, stats, tok, vi, vo, gl, done, -1, packed=Wp)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / iters
print('proj_gumbel packed=%d M=%d N=%d K=%d greedy=%d z=%d: %.1f us per call (kernel + finalize), %.1f TFLOP/s' %
      (packed, M, N, K, greedy, write_z, us, 2.0 * M * N * K / us / 1e6))
