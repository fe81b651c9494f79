#!/usr/bin/env python3
"""Sweep times at chosen batch sizes (GPU box): usage lstm_bench_b.py H B [B ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd
from a4nt_amd import ops
T = 100
H = int(sys.argv[1])
for B in [int(b) for b in sys.argv[2:]]:
    mt, g = ops.lstm_geometry(B, H)
    w_hh = torch.randn(4 * H, H, device='cuda') * 0.04
    gx0 = torch.randn(T, B, 4 * H, device='cuda')
    dhs = torch.randn(T, B, H, device='cuda')
    res = {}
    for name in ('fwd', 'bwd'):
        ts = []
        for it in range(5):
            gx = gx0.clone()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if name == 'fwd':
                e0.record(); hs, cs, hN, cN = ops.lstm_seq_fwd(gx, w_hh, None); e1.record()
                keep = (gx, cs)
            else:
                e0.record(); ops.lstm_seq_bwd(w_hh, keep[0], keep[1], None, None, dhs, None, None, False); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[name] = min(ts)
    ops.check_timeouts()
    print('H=%d B=%d mt=%d groups=%d: fwd %.3f ms (%.2f us/step)  bwd %.3f ms (%.2f us/step)' % (
        H, B, mt, g, res['fwd'], 10 * res['fwd'], res['bwd'], 10 * res['bwd']))
