#!/bin/bash
# team kernel stamps on the c5s workload: team size variants
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_team3; mkdir -p $OUT
D=/tmp/wfa_ts; mkdir -p $D/wfa_amd/lib && cp wfa_amd/*.py $D/wfa_amd/
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWFA_TEAM_STAMPS -shared -o $D/wfa_amd/lib/libwfahip.so wfa_amd/csrc/wfa_host.hip wfa_amd/csrc/wfa_gen.cpp wfa_amd/csrc/wfa_multi.cpp 2>/dev/null
cd $D && timeout 600 python3 - $D "$@" > $REPO/$OUT/stamps.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import wfa_amd as w
data = w.generate_pairs(5, 8, 100000, 0.10, n_threads=8)
for spec in sys.argv[2:] or ["team_wgs=0"]:
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    for kv in spec.split(","):
        k, v = kv.split("="); al.set_option(k, int(v))
    for rep in range(3):
        t0 = time.time(); r = al.align_arrays(*data); print(spec, "wall", time.time() - t0, al.last_timing(), flush=True)
    al.close()
PY
cd $REPO; grep "band ends\|wall 0\|inside P1\|us: P1" $OUT/stamps.txt | tail -10 | cut -c1-260
