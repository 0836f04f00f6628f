#!/bin/bash
# Build a named library variant for scripts/ab.sh: the duo unit (and, with HOST=1, the three host-side units) rebuilt with extra flags,
# every other object taken from build/obj (run `make` first).  Usage: [HOST=1] scripts/mkvariant.sh <name> [extra hipcc flags...]
set -e
N=$1; shift
O=build/variants/obj/$N; mkdir -p $O
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
hipcc $F "$@" -mllvm -amdgpu-atomic-optimizer-strategy=None -c -o $O/wfa_duo.o wfa_amd/csrc/wfa_duo.hip &
if [ "${HOST:-0}" = 1 ]; then for u in wfa_host wfa_entry wfa_debug; do hipcc $F "$@" -c -o $O/$u.o wfa_amd/csrc/$u.hip & done; fi
wait
OBJS=""
for o in build/obj/*.o; do b=$(basename $o); if [ -f $O/$b ]; then OBJS="$OBJS $O/$b"; else OBJS="$OBJS $o"; fi; done
hipcc -fPIC --offload-arch=gfx950 -shared -o build/variants/$N.so $OBJS
echo built build/variants/$N.so
