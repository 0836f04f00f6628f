#!/bin/bash
# team kernel stamps on the c5s workload, current ladder (8 teams), TEAM_U 2 / 4
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_team1; mkdir -p $OUT
for U in 2 4; do
D=/tmp/wfa_ts$U; mkdir -p $D/wfa_amd/lib && cp wfa_amd/*.py $D/wfa_amd/
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWFA_TEAM_STAMPS -DWFA_TEAM_U=$U -shared -o $D/wfa_amd/lib/libwfahip.so wfa_amd/csrc/wfa_host.hip wfa_amd/csrc/wfa_gen.cpp wfa_amd/csrc/wfa_multi.cpp 2>/dev/null
cd $D && timeout 300 python3 - $D > $REPO/$OUT/team_stamps_u$U.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import wfa_amd as w
data = w.generate_pairs(5, 8, 100000, 0.10, n_threads=8)
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
for rep in range(3):
    t0 = time.time(); r = al.align_arrays(*data); print("wall", time.time() - t0, al.last_timing(), flush=True)
PY
cd $REPO; echo "== U=$U"; tail -12 $OUT/team_stamps_u$U.txt | cut -c1-260
done
