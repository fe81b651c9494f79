#!/usr/bin/env python3
"""Per-step GPU time budget from a rocprofv3 kernel trace (rocpd sqlite database).

  python tools/trace_step.py <results.db> [marker]

A kernel that runs exactly once per training step (default: proj_gumbel_pack_kernel, the per-generation repack of
the output projection) marks the step boundaries.  For the steps between the first and the last marker: wall time
per step (marker to marker), the sum of kernel durations, the idle remainder, and the kernels ranked by time per
step.  Box-to-box clocks differ by a few percent: compare two builds from traces taken in ONE gpurun call."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'^_Z\d+', '', name)
    return name[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else 'proj_gumbel_pack_kernel'
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = cur.execute('select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start'
                       % (kd, ks)).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        sys.exit('marker %r found %d times: need at least 3' % (marker, len(marks)))
    marks = marks[1:]                                  # the first step includes warm-up effects
    nsteps = len(marks) - 1
    seg = rows[marks[0]:marks[-1]]
    wall = (rows[marks[-1]][0] - rows[marks[0]][0]) / 1e6 / nsteps
    by = defaultdict(lambda: [0, 0.0])
    busy = 0.0
    for st, en, name in seg:
        by[short(name)][0] += 1
        by[short(name)][1] += (en - st) / 1e6
        busy += (en - st) / 1e6
    busy /= nsteps
    print('%d steps: wall %.2f ms/step, kernels %.2f ms/step, idle %.2f ms/step, %d launches/step'
          % (nsteps, wall, busy, wall - busy, len(seg) // nsteps))
    for name, (n, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])[:28]:
        print('  %-70s %6.1f /step  %7.3f ms/step  %7.1f us avg' % (name, n / nsteps, ms / nsteps, 1e3 * ms / n))


if __name__ == '__main__':
    main()
