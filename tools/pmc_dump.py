#!/usr/bin/env python3
"""Sum every counter of a rocprofv3 --pmc pass per (kernel name, grid size).  usage: pmc_dump.py <dir> [name filter]"""
import csv, glob, os, sys
from collections import defaultdict
acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(set)
for fn in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(fn, newline='')):
        if len(sys.argv) > 2 and sys.argv[2] not in row['Kernel_Name']:
            continue
        key = (row['Kernel_Name'][:60], row['Grid_Size'])
        acc[key][row['Counter_Name']] += float(row['Counter_Value'])
        cnt[key].add(row['Dispatch_Id'])
for key in sorted(acc):
    print(key, 'launches', len(cnt[key]))
    c = acc[key]
    for k in sorted(c):
        print('    %-28s %.4g' % (k, c[k]))
    if 'SQ_WAVE_CYCLES' in c:
        w = c['SQ_WAVE_CYCLES']
        print('    -> of wave cycles: parked (waitcnt/barrier) %.0f %%, issue-stalled %.0f %%, issuing %.0f %%' % (
            100 * c.get('SQ_WAIT_ANY', 0) / w, 100 * c.get('SQ_WAIT_INST_ANY', 0) / w, 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / w))
