#!/usr/bin/env python3
"""Per-kernel summary (the columns of rocprofv3 --stats' kernel_stats.csv) from a rocprofv3 rocpd database.
usage: rocpd_stats.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = {}
    for name, start, end in cur.execute("select %s, start, end from kernels" % name_col):
        rows.setdefault(name, []).append(end - start)
    total = float(sum(sum(v) for v in rows.values()))
    out = []
    for name, v in rows.items():
        n, s = len(v), float(sum(v))
        mean = s / n
        sd = (sum((x - mean) ** 2 for x in v) / n) ** 0.5
        out.append((name, n, int(s), mean, 100.0 * s / total, min(v), max(v), sd))
    out.sort(key=lambda r: -r[2])
    w = csv.writer(open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in out:
        w.writerow([r[0], r[1], r[2], round(r[3], 3), round(r[4], 2), r[5], r[6], round(r[7], 3)])


if __name__ == '__main__':
    main()
