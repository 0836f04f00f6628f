#!/bin/bash
# Build a named library variant for scripts/ab.sh.  Usage: scripts/mkvariant.sh <name> [extra hipcc flags...]
set -e
N=$1; shift
O=build/variants/obj/$N; mkdir -p $O
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
hipcc $F "$@" -mllvm -amdgpu-atomic-optimizer-strategy=None -c -o $O/duo.o wfa_amd/csrc/wfa_duo.hip 2>/dev/null &
hipcc $F "$@" -c -o $O/host.o wfa_amd/csrc/wfa_host.hip 2>/dev/null &
hipcc $F "$@" -c -o $O/gen.o wfa_amd/csrc/wfa_gen.cpp 2>/dev/null
hipcc $F "$@" -c -o $O/multi.o wfa_amd/csrc/wfa_multi.cpp 2>/dev/null
wait
hipcc -fPIC --offload-arch=gfx950 -shared -o build/variants/$N.so $O/host.o $O/gen.o $O/multi.o $O/duo.o
echo built build/variants/$N.so
