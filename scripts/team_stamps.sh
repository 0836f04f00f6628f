#!/bin/bash
# Diagnostic: build a copy of the library with wfa_team_kernel's in-kernel stamps (-DWFA_TEAM_STAMPS) and run the configs[4]
# sample (8 x 100 kbp @10 %, semi-global, wf-adaptive) with it; the per-team phase table goes to stderr.  Never used for timing
# numbers.  Usage (GPU box): scripts/team_stamps.sh [pairs] [opts k=v,k=v]      WFA_EXTRA = extra hipcc flags.
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
D=/tmp/wfa_tstamps
mkdir -p $D/wfa_amd/lib
cp -r wfa_amd/*.py $D/wfa_amd/
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWFA_TEAM_STAMPS $WFA_EXTRA"
# (only the units the macro changes are rebuilt: the router and the long-pair kernels; the rest are the in-tree objects)
make -s -j8 wfa_amd/lib/libwfahip.so
for f in wfa_host.hip wfa_long.hip; do hipcc $F -c -o $D/${f%.*}.o wfa_amd/csrc/$f & done
wait
OBJS=$(ls build/obj/*.o | grep -v "wfa_host.o\|wfa_long.o")
hipcc -fPIC --offload-arch=gfx950 -shared -o $D/wfa_amd/lib/libwfahip.so $D/wfa_host.o $D/wfa_long.o $OBJS
cd $D && WFAHIP_DEBUG_TIMING=1 python3 - "$@" <<'PY'
import sys, time
sys.path.insert(0, "/tmp/wfa_tstamps")
import wfa_amd as w
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
data = w.generate_pairs(5, n, 100000, 0.10, n_threads=32)
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False))
al.AdaptiveReduction(w.DefaultAdaptiveOption)
for kv in filter(None, (sys.argv[2] if len(sys.argv) > 2 else "").split(",")):
    k, v = kv.split("="); al.set_option(k, int(v))
r = al.align_arrays(*data)
t0 = time.perf_counter(); r = al.align_arrays(*data); t1 = time.perf_counter()
print("wall", t1 - t0, al.last_timing())
PY
