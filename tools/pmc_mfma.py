#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from one rocprofv3 counter pass over bench.py.

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- \\
      python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-rooflines
  python3 tools/pmc_mfma.py gpurun_out/pmc_mfma profiles/r01_pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES counts the cycles the matrix pipes are busy, summed over the chip's 1024 SIMDs (32 per
v_mfma_f32_16x16x4_f32, 64 per 32x32x2_f32); GRBM_GUI_ACTIVE is the busy time of the dispatch in cycles, summed over
the 8 XCDs (MI355X_MICROARCH.md, DVFS section).  utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

NAMES = [(r'sgemm_kernel<false, false, 2, 2, \d, 1, true', 'sgemm_nn_128_nll'),
         (r'sgemm_kernel<false, true, 2', 'sgemm_nt_128'), (r'sgemm_kernel<true, false, 2, 2, \d, 1, false, true', 'sgemm_tn_128_colsum'),
         (r'sgemm_kernel<true, false, 2', 'sgemm_tn_128'),
         (r'sgemm_kernel<false, false, 2', 'sgemm_nn_128'), (r'sgemm_kernel<.*, 1, 1, 4', 'sgemm_64'),
         (r'lstm_fwd2_kernel', 'lstm_fwd_pair'), (r'lstm_bwdx2_kernel|lstm_bwd2_kernel', 'lstm_bwd_pair'),
         (r'lstm_dec_kernel', 'lstm_dec_step'), (r'lstm_fwd_kernel', 'lstm_fwd'),
         (r'lstm_bwdx_kernel|lstm_bwd_kernel', 'lstm_bwd'), (r'conv1d_kernel', 'conv1d'),
         (r'proj_gumbel_kernel', 'proj_gumbel'), (r'nt_rows_kernel', 'sgemm_nt_rows'), (r'skinny_kernel', 'sgemm_skinny')]


def main(d, out):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no counter_collection.csv under %s' % d)
    for fn in files:
        with open(fn, newline='') as f:
            for row in csv.DictReader(f):
                lab = next((l for p, l in NAMES if re.search(p, row['Kernel_Name'])), None)
                if lab is None:
                    continue
                acc[lab][row['Counter_Name']] += float(row['Counter_Value'])
                if row['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                    n[lab] += 1
    res = {}
    for lab, c in sorted(acc.items()):
        busy, act = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), c.get('GRBM_GUI_ACTIVE', 0.0)
        if act <= 0:
            continue
        res[lab] = {'launches': n[lab], 'mfma_busy_cycles': busy, 'gui_active_cycles_sum_xcd': act,
                    'mfma_util': busy / (act / 8.0 * 1024.0)}
        print('%-38s launches %5d  MFMA utilisation %5.1f %%' % (lab, n[lab], 100 * res[lab]['mfma_util']))
    with open(out, 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
