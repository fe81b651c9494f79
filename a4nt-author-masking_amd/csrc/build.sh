#!/bin/bash
# Builds liba4nt_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [--clean]
# --clean: compile every source again instead of reusing objects newer than their sources.
set -e
cd "$(dirname "$0")"
ARCH=gfx950
if [ "$1" = "--clean" ]; then rm -f ./*.o ../liba4nt_hip.so; fi
FLAGS="-O3 -std=c++17 --offload-arch=$ARCH -fPIC -fno-gpu-rdc -Wno-unused-result"
OBJS=""
pids=()
for f in capi sgemm skinny ntrows lstm rows misc projgumbel conv1d genloop genpersist optim; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ gemm_stage.h -nt $f.o ] || [ pg_epilogue.h -nt $f.o ] || [ ../../include/a4nt_hip.h -nt $f.o ]; then
    hipcc $FLAGS -c $f.hip -o $f.o &
    pids+=($!)
  fi
  OBJS="$OBJS $f.o"
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=$ARCH -shared -fPIC -o ../liba4nt_hip.so $OBJS
echo "built $(cd .. && pwd)/liba4nt_hip.so"
