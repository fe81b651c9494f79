#!/usr/bin/env python3
"""Sequence of the kernels of ONE training step from a rocprofv3 kernel trace (csv): every dispatch outside the two
decoding loops with its offset, duration and the idle gap in front of it; the loops are folded into one line each.

  python tools/trace_seq.py <..._kernel_trace.csv> [step index]"""
import csv
import re
import sys

LOOP = ('proj_gumbel_kernel', 'lstm_dec_kernel', 'nt_rows_kernel', 'gumbel_bwd', 'lstm_cell_bwd4', 'slab_reduce4',
        'sgemm_kernel<false, true, 1, 1', 'sgemm_kernel<false, false, 1, 1')


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'at::native::', '', n)
    return n[:100]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'proj_gumbel_pack_kernel' in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
seg = rows[marks[k]:marks[k + 1]]
t0 = seg[0][0]
prev_end = t0
fold = None
tot_small = 0.0
for st, en, name in seg:
    inloop = any(x in name for x in LOOP)
    if inloop:
        if fold is None:
            fold = [st, en, 1, en - st]
        else:
            fold[1] = en; fold[2] += 1; fold[3] += en - st
        prev_end = en
        continue
    if fold is not None:
        print('%9.1f us  [loop: %d launches, %.1f us of kernels, %.1f us wall]' % ((fold[0] - t0) / 1e3, fold[2], fold[3] / 1e3, (fold[1] - fold[0]) / 1e3))
        fold = None
    d = (en - st) / 1e3
    if d < 30:
        tot_small += d
    print('%9.1f us  %7.1f us  gap %5.1f  %s' % ((st - t0) / 1e3, d, (st - prev_end) / 1e3, short(name)))
    prev_end = max(prev_end, en)
print('step wall %.2f ms; kernels under 30 us outside the loops: %.2f ms' % ((seg[-1][1] - t0) / 1e6, tot_small / 1e3))
