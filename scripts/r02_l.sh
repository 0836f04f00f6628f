#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_l; mkdir -p $OUT
cp wfa_amd/lib/libwfahip.so /tmp/orig.so
for v in s4 s5; do cp build/variants/$v.so wfa_amd/lib/libwfahip.so; echo "== $v"; timeout 600 python scripts/chunks_ab.py 2>&1 | tail -4; done > $OUT/chunks.txt 2>&1
cp /tmp/orig.so wfa_amd/lib/libwfahip.so
cat $OUT/chunks.txt
