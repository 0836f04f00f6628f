#!/bin/bash
# Build a named library variant for scripts/ab.sh.  Usage: scripts/mkvariant.sh <name> [extra hipcc flags...]
set -e
N=$1; shift
mkdir -p build/variants
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -shared -o build/variants/$N.so wfa_amd/csrc/wfa_host.hip wfa_amd/csrc/wfa_gen.cpp wfa_amd/csrc/wfa_multi.cpp 2>/dev/null
echo built build/variants/$N.so
