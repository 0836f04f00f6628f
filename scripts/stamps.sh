#!/bin/bash
export WFAHIP_DEBUG=1  # (the option knobs below are debug / experiment knobs)
# Diagnostic (GPU box): one pass of the headline workload with the stamped library build/variants/stamps.so
# (HOST=1 scripts/mkvariant.sh stamps -DWFA_STAMPS): per-phase s_memtime shares + event counts of the forward kernel on
# stderr.  Never used for timing numbers.  Usage: scripts/stamps.sh [pairs [length [error]]]   WFA_OPTS=key=value,...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
V=${VARIANT:-stamps}
rm -rf /tmp/wfa_stamps; mkdir -p /tmp/wfa_stamps/wfa_amd/lib
cp -r wfa_amd/*.py /tmp/wfa_stamps/wfa_amd/
cp build/variants/$V.so /tmp/wfa_stamps/wfa_amd/lib/libwfahip.so
cd /tmp/wfa_stamps && python3 - "$@" <<'PY'
import sys, os, numpy as np
sys.path.insert(0, "/tmp/wfa_stamps")
import wfa_amd as w
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
err = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
data = w.generate_pairs(3, n, length, err, n_threads=32)
al = w.New()
if length >= 500: al.AdaptiveReduction(w.DefaultAdaptiveOption)
for kv in filter(None, os.environ.get("WFA_OPTS", "").split(",")):  # e.g. WFA_OPTS=duo=1,blk_wide=0
    k, v = kv.split("="); al.set_option(k, int(v))
r = al.align_arrays(*data); r = al.align_arrays(*data)
print(al.last_timing())
PY
