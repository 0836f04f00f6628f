#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_team8; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "config5 or team or long or semi" --durations=6 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
# the same stale-word test with the release taken out of the barriers: expected to FAIL (it shows the test sees the race)
WFA_TEST_OPTS=team_strict=0 timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "stale_arena" > $OUT/pytest_nostrict.log 2>&1; echo "team_strict=0: pytest rc $?" | tee -a $OUT/pytest_nostrict.log
tail -5 $OUT/pytest_nostrict.log | cut -c1-200
bash scripts/r02_team7.sh 30 900 team_strict=1,arena_poison=1
