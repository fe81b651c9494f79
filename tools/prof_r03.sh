#!/bin/bash
# Round-3 profile set (GPU box): kernel trace of a bench run + the three counter passes, each in a run of its own.
# usage: tools/prof_r03.sh [workload (adv | cfg5)] [tag]
# Outputs under gpurun_out/r03p_<tag>/; reduce with tools/rocpd_stats.py, tools/pmc_mfma.py, tools/pmc_traffic.py.
set -e
W=${1:-adv}
TAG=${2:-$W}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p_$TAG
mkdir -p $O
STEPS=10; [ "$W" = cfg5 ] && STEPS=4
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --workload $W --steps $STEPS --warmup 3 > $O/trace.log 2>&1
B="python3 $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-extra-rooflines"
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- $B > $O/pmc_mfma.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
echo done
