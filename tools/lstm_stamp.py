#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the persistent forward sweep (workgroup 0, wave 0).
Build the stamp variant first (csrc: hipcc -DA4NT_STAMP lstm.hip -> liba4nt_hip_stamp.so) and run with
A4NT_HIP_LIB pointing at it.  Shares only: the stamp build forbids overlaps the product build has."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd
from a4nt_amd import ops
T, H = 100, 512
names = ['poll+barrier', 'A loads landed', 'MFMA+LDS write', 'barrier(e)', 'cell update+stores issue', 'barrier(f)',
         'publish+drain', 'barrier+signal']
for B in (128, 256):
    w_hh = torch.randn(4 * H, H, device='cuda') * 0.04
    gx0 = torch.randn(T, B, 4 * H, device='cuda')
    for it in range(3):
        gx = gx0.clone()
        ops.lstm_seq_fwd(gx, w_hh, None)
    torch.cuda.synchronize()
    xch, sync, g = ops._lstm_ws(gx.device, B, H, False)
    words = sync[32 * g + 8: 32 * g + 24].cpu().numpy().astype('uint32')
    vals = [int(words[2 * k]) | (int(words[2 * k + 1]) << 32) for k in range(8)]
    tot = sum(vals)
    print('B=%d  total %.0f cycles/step (100 MHz ticks? see below)' % (B, tot / T))
    for n, v in zip(names, vals):
        print('   %-26s %8.1f per step  %5.1f%%' % (n, v / T, 100.0 * v / tot))

# (The r03 backward sweep, lstm_bwdx_body, carries no phase stamps: it was tuned with ablation builds instead -- one part
# of the step switched off per build, timed with tools/lstm_bench_b.py; DESIGN.md section 4.1.  The 1-D backward kernel
# that serves H <= 128 still has them.)
