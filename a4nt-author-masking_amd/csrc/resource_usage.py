#!/usr/bin/env python3
"""Print VGPR/AGPR/SGPR/LDS/spill per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
here = os.path.dirname(os.path.abspath(__file__))
for f in sys.argv[1:] or ['sgemm', 'lstm', 'rows', 'misc']:
    out = subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-c', os.path.join(here, f + '.hip'),
                          '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
        if not m:
            continue
        kv = m.group(1)
        if kv.startswith('Function Name') or kv.startswith('Name'):
            if cur:
                print(cur)
            cur = {'k': kv.split(':', 1)[1].strip()[:60]}
        else:
            k, _, v = kv.partition(':')
            if k.strip() in ('VGPRs', 'AGPRs', 'TotalSGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]',
                             'SGPRs Spill', 'VGPRs Spill', 'LDS Size [bytes/block]'):
                cur[k.strip().split(' ')[0] + ('Spill' if 'Spill' in k else '')] = v.strip()
    if cur:
        print(cur)
