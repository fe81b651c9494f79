#!/usr/bin/env python3
"""Per-step wall time of the benchmark's adversarial step over a long run (does the step time drift?)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.cuda.set_device(0)
import a4nt_amd as impl
torch.manual_seed(0)
G = impl.CharTranslator(bench.gen_params(bench.B)).train()
D = impl.CharLstm(bench.eval_params()).train()
optG, optD = impl.make_optimizers(G, D)
batches = bench.make_batches(impl, 1234, bench.B, 6)
ts, hs = [], []
for it in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.adv_step(impl, G, D, optG, optD, batches, it, bench.B, (None, None))
    hs.append((time.perf_counter() - t0) * 1e3)          # the host has queued the whole step (no sync inside it)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print('ms per step:', ' '.join('%.1f' % t for t in ts))
print('host ms per step (enqueue only):', ' '.join('%.1f' % t for t in hs))
print('mean first 10 after 3: %.2f, last 10: %.2f' % (sum(ts[3:13]) / 10, sum(ts[-10:]) / 10))
print('mem allocated GB %.2f reserved %.2f' % (torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
