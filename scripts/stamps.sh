#!/bin/bash
# Diagnostic: build a stamped copy of the library (per-phase s_memtime shares of the packed kernel) and run
# one pass of the bench workload with it.  Never used for timing numbers.  WFA_EXTRA = extra hipcc flags (experiments).
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p /tmp/wfa_stamps/wfa_amd/lib
cp -r wfa_amd/*.py /tmp/wfa_stamps/wfa_amd/
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWFA_STAMPS $WFA_EXTRA"
hipcc $F -mllvm -amdgpu-atomic-optimizer-strategy=None -c -o /tmp/wfa_stamps/duo.o wfa_amd/csrc/wfa_duo.hip
for f in wfa_host.hip wfa_gen.cpp wfa_multi.cpp; do hipcc $F -c -o /tmp/wfa_stamps/${f%.*}.o wfa_amd/csrc/$f; done
hipcc -fPIC --offload-arch=gfx950 -shared -o /tmp/wfa_stamps/wfa_amd/lib/libwfahip.so /tmp/wfa_stamps/wfa_host.o /tmp/wfa_stamps/wfa_gen.o /tmp/wfa_stamps/wfa_multi.o /tmp/wfa_stamps/duo.o
cd /tmp/wfa_stamps && python3 - "$@" <<'PY'
import sys, numpy as np
sys.path.insert(0, "/tmp/wfa_stamps")
import wfa_amd as w
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
err = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
data = w.generate_pairs(3, n, length, err, n_threads=32)
al = w.New()
if length >= 500: al.AdaptiveReduction(w.DefaultAdaptiveOption)
import os
for kv in filter(None, os.environ.get("WFA_OPTS", "").split(",")):  # e.g. WFA_OPTS=duo=1,blk_wide=0
    k, v = kv.split("="); al.set_option(k, int(v))
r = al.align_arrays(*data); r = al.align_arrays(*data)
print(al.last_timing())
PY
