#!/usr/bin/env python3
"""Condensed view of bench.py's JSON line (stdin)."""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.2f ms/step  %.0f %s' % (d['ms_per_step'], d['value'], d['unit']))
r = d['roofline']
for k in [r] + r.get('other_kernels', []):
    print('  %-14s %6.1f TF  %5.1f%%  %8.3f ms avg  x%d' % (k['kernel'], k['achieved'], 100 * k['frac'], k['avg_launch_ms'], k['launches']))
if 'cpu_baseline' in d:
    print('  cpu', d['cpu_baseline'])
