#!/bin/bash
# A/B harness for the team kernel: run scripts/team_sweep.py with every library under build/variants/.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
cp wfa_amd/lib/libwfahip.so /tmp/libwfahip.orig.so
for v in build/variants/*.so; do cp $v wfa_amd/lib/libwfahip.so; echo "== $(basename $v)"; python scripts/team_sweep.py "$@" 2>&1 | tail -3; done
cp /tmp/libwfahip.orig.so wfa_amd/lib/libwfahip.so
