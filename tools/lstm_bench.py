#!/usr/bin/env python3
"""Micro-benchmark of the persistent LSTM sweeps (GPU box only)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd
from a4nt_amd import ops, _lib
lib = _lib.load()
T = 100
H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for maxwg in (256, 512):
    lib.a4nt_lstm_set_max_workgroups(maxwg)
    ops._ws.clear()
    for B in ((128, 256, 512) if maxwg == 256 else (256,)):
        try:
            mt, g = ops.lstm_geometry(B, H)
        except Exception as e:
            print('B=%d: %s' % (B, e)); continue
        w_hh = torch.randn(4 * H, H, device='cuda') * 0.04
        gx0 = torch.randn(T, B, 4 * H, device='cuda')
        dhs = torch.randn(T, B, H, device='cuda')
        res = {}
        for name in ('fwd', 'bwd'):
            ts = []
            for it in range(6):
                gx = gx0.clone()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if name == 'fwd':
                    e0.record(); hs, cs, hN, cN = ops.lstm_seq_fwd(gx, w_hh, None); e1.record()
                    keep = (gx, cs)
                else:
                    hs, cs, hN, cN = ops.lstm_seq_fwd(gx, w_hh, None)
                    e0.record(); ops.lstm_seq_bwd(w_hh, gx, cs, None, None, dhs, None, None, False); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ops.check_timeouts()
            ms = sorted(ts)[len(ts) // 2]
            res[name] = ms
        fl = 8.0 * T * B * H * H
        print('maxwg=%d B=%3d mt=%d groups=%2d wgs=%3d  fwd %.3f ms (%.2f us/step, %.1f TF, %.0f%%)  bwd %.3f ms (%.2f us/step, %.1f TF, %.0f%%)' % (
            maxwg, B, mt, g, g * H // 16, res['fwd'], res['fwd'] * 1e3 / T, fl / res['fwd'] / 1e9, fl / res['fwd'] / 1e9 / 1.573,
            res['bwd'], res['bwd'] * 1e3 / T, fl / res['bwd'] / 1e9, fl / res['bwd'] / 1e9 / 1.573))
lib.a4nt_lstm_set_max_workgroups(256)
