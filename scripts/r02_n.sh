#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r02_n; mkdir -p $OUT
mkdir -p /tmp/wfa_stamps/wfa_amd/lib
cp -r wfa_amd/*.py /tmp/wfa_stamps/wfa_amd/
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DWFA_STAMPS -shared -o /tmp/wfa_stamps/wfa_amd/lib/libwfahip.so wfa_amd/csrc/wfa_host.hip wfa_amd/csrc/wfa_gen.cpp wfa_amd/csrc/wfa_multi.cpp 2>/dev/null
cd /tmp/wfa_stamps && WFAHIP_NO_UPLOAD_OVERLAP=1 python3 - > $OUT/stamps.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "/tmp/wfa_stamps")
import wfa_amd as w
data = w.generate_pairs(3, 1000000, 1000, 0.05, n_threads=32)
for nl in (0, 1):
    al = w.New(); al.AdaptiveReduction(w.DefaultAdaptiveOption); al.set_option("narrow_long", nl)
    print("==== narrow_long", nl, flush=True)
    r = al.align_arrays(*data); r = al.align_arrays(*data)
    print(al.last_timing(), flush=True)
    al.close()
PY
cat $OUT/stamps.txt | cut -c1-260
