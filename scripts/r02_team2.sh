#!/bin/bash
# wave mode of the team kernel: parity tests, then stamps on the c5s workload
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_team2; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "team or config5 or semi or long or generic or dump" --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for U in 2; do
D=/tmp/wfa_ts$U; mkdir -p $D/wfa_amd/lib && cp wfa_amd/*.py $D/wfa_amd/
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWFA_TEAM_STAMPS -DWFA_TEAM_U=$U -shared -o $D/wfa_amd/lib/libwfahip.so wfa_amd/csrc/wfa_host.hip wfa_amd/csrc/wfa_gen.cpp wfa_amd/csrc/wfa_multi.cpp 2>/dev/null
cd $D && timeout 300 python3 - $D > $REPO/$OUT/team_stamps_u$U.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import wfa_amd as w
data = w.generate_pairs(5, 8, 100000, 0.10, n_threads=8)
for wave in (1, 0):
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    al.set_option("team_wave", wave)
    for rep in range(3):
        t0 = time.time(); r = al.align_arrays(*data); print("wave", wave, "wall", time.time() - t0, al.last_timing(), flush=True)
    al.close()
PY
cd $REPO; echo "== U=$U"; tail -22 $OUT/team_stamps_u$U.txt | cut -c1-330
done
timeout 600 python bench.py --config c5s --steps 4 --warmup 1 > $OUT/bench_c5s.json 2> $OUT/bench_c5s.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_team2/bench_c5s.json").read().strip().splitlines()[-1]); print("c5s", d["value"], d["ms_per_step"])
PY
