#!/bin/bash
# Diagnostic build of the library with the persistent generation kernel's phase stamps (-DGP_STAMP) -> exp/liba4nt_GPSTAMP.so
# (run tools/gen_fwd_time.py with A4NT_HIP_LIB pointing at it).  $1: extra -D flags for ablations.
set -e
cd "$(dirname "$0")/../a4nt-author-masking_amd/csrc"
./build.sh > /dev/null
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wno-unused-result -DGP_STAMP $1 -c genpersist.hip -o /tmp/genpersist_stamp.o
OBJS=$(ls *.o | grep -v genpersist.o)
mkdir -p ../../exp
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../exp/liba4nt_GPSTAMP.so $OBJS /tmp/genpersist_stamp.o
echo built exp/liba4nt_GPSTAMP.so
