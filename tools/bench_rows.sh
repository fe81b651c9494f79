#!/bin/bash
# The other rows of DESIGN.md section 6 (GPU box), one bench run each, bench lines into gpurun_out/rows/.
set -e
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/rows
mkdir -p $O
F="--no-cpu-baseline --no-extra-rooflines"
timeout -k 10 200 python $R/bench.py $F > $O/adv.log 2>&1
timeout -k 10 200 python $R/bench.py $F --lstm-workgroups 256 > $O/adv_wg256.log 2>&1
timeout -k 10 200 python $R/bench.py $F --ragged > $O/adv_ragged.log 2>&1
timeout -k 10 200 python $R/bench.py $F --separate-gen > $O/adv_sepgen.log 2>&1
timeout -k 10 200 python $R/bench.py $F --workload cfg1 > $O/cfg1.log 2>&1
timeout -k 10 200 python $R/bench.py $F --workload cfg2 > $O/cfg2.log 2>&1
timeout -k 10 300 python $R/bench.py $F --workload cfg5 --steps 10 --warmup 2 > $O/cfg5.log 2>&1
for f in adv adv_wg256 adv_ragged adv_sepgen cfg1 cfg2 cfg5; do echo -n "$f: "; tail -1 $O/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['unit'])"; done
