#!/bin/bash
# team kernel: build variants with -D flags and run the c5s sample 4 times each (determinism + stamps)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_team4; mkdir -p $OUT
i=0
for FLAGS in "$@"; do
i=$((i+1)); D=/tmp/wfa_tv$i; mkdir -p $D/wfa_amd/lib && cp wfa_amd/*.py $D/wfa_amd/
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWFA_TEAM_STAMPS $FLAGS -shared -o $D/wfa_amd/lib/libwfahip.so wfa_amd/csrc/wfa_host.hip wfa_amd/csrc/wfa_gen.cpp wfa_amd/csrc/wfa_multi.cpp 2>/dev/null
cd $D && timeout 600 python3 - $D > $REPO/$OUT/stamps_$i.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import wfa_amd as w
data = w.generate_pairs(5, 8, 100000, 0.10, n_threads=8)
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
for rep in range(5):
    t0 = time.time(); r = al.align_arrays(*data); t = al.last_timing(); print("wall %.4f kernel_ms %.1f cells %d ops %d score_sum %d" % (time.time() - t0, t.kernel_ms, t.cells_stored, t.ops_written, int(r.score.sum())), flush=True)
al.close()
PY
cd $REPO; echo "== $FLAGS"; grep "^wall" $OUT/stamps_$i.txt; grep "team 0" $OUT/stamps_$i.txt | tail -1 | cut -c1-250
done
