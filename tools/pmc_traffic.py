#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc passes over bench.py to HBM bytes per launch, per kernel.

Usage (on the GPU box, each counter in a run of its own, no trace flags):
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-rooflines
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...same...
  python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled (every read stream of
these kernels is a 16-byte-per-lane buffer load); WRITE_SIZE is exact for 16-byte streaming stores.
Working sets under 256 MiB can be served from the Infinity Cache and still be counted, so "traffic"
here is memory-side requests, an upper bound on DRAM bytes.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

LABELS = [
    (r'sgemm_kernel<(true|false), (true|false), (\d), \d, \d(?:, \d)?(?:, (true|false))?(?:, (true|false))?>', None),
    (r'lstm_fwd2_kernel', 'lstm_seq_fwd_pair'),
    (r'lstm_bwdx2_kernel|lstm_bwd2_kernel', 'lstm_seq_bwd_pair'),
    (r'nt_rows_kernel', 'sgemm_nt_rows'),
    (r'lstm_dec_kernel', 'lstm_dec_step'),
    (r'lstm_fwd_kernel', 'lstm_seq_fwd'),
    (r'lstm_bwdx_kernel|lstm_bwd_kernel', 'lstm_seq_bwd'),
    (r'nll_grad_rows_kernel', 'nll_grad_rows'),
    (r'conv1d_kernel', 'conv1d'),
    (r'proj_gumbel_kernel', 'proj_gumbel'),
    (r'proj_gumbel_final_kernel', 'proj_gumbel_final'),
    (r'gumbel_bwd(_reg)?_kernel', 'gumbel_bwd'),
    (r'nll_rows_kernel', 'nll_rows'),
    (r'lstm_cell_bwd4?_kernel', 'lstm_cell_bwd'),
    (r'slab_reduce4?_kernel', 'slab_reduce'),
    (r'colsum_kernel', 'colsum'),
]


def label_of(name):
    m = re.search(LABELS[0][0], name)
    if m:
        ta, tb, w = m.group(1) == 'true', m.group(2) == 'true', int(m.group(3))
        nll = '_nll' if m.group(4) == 'true' else ''        # 7th template argument: NLL epilogue; 8th: column sums of B
        nll += '_colsum' if m.group(5) == 'true' else ''
        return 'sgemm_%s_%d%s' % ('tn' if ta else ('nt' if tb else 'nn'), 128 if w == 2 else 64, nll)
    for pat, lab in LABELS[1:]:
        if re.search(pat, name):
            return lab
    return None


def collect(d, counter):
    """{label: [sum KiB, launches]} from every *counter_collection.csv under d."""
    acc = defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no counter_collection.csv under %s' % d)
    for fn in files:
        with open(fn, newline='') as f:
            for row in csv.DictReader(f):
                if row['Counter_Name'] != counter:
                    continue
                lab = label_of(row['Kernel_Name'])
                if lab is None:
                    continue
                acc[lab][0] += float(row['Counter_Value'])
                acc[lab][1] += 1
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fe, wr = collect(fetch_dir, 'FETCH_SIZE'), collect(write_dir, 'WRITE_SIZE')
    kernels = {}
    for lab in sorted(set(fe) | set(wr)):
        fk, fn_ = fe.get(lab, [0.0, 0])
        wk, wn = wr.get(lab, [0.0, 0])
        rd = 2.0 * 1024.0 * fk / max(1, fn_)
        wb = 1024.0 * wk / max(1, wn)
        kernels[lab] = {'launches': fn_, 'fetch_bytes_per_launch': rd, 'write_bytes_per_launch': wb,
                        'hbm_bytes_per_launch': rd + wb}
    with open(out, 'w') as f:
        json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) over bench.py --steps 1 --warmup 1',
                   'corrections': 'KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read undercount)', 'kernels': kernels},
                  f, indent=1, sort_keys=True)
    for lab, e in kernels.items():
        print('%-20s launches %5d  read %10.3f MB  write %10.3f MB' % (lab, e['launches'], e['fetch_bytes_per_launch'] / 1e6,
                                                                    e['write_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()
