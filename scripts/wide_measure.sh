#!/bin/bash
# wfa_wide_kernel on the GPU box: its parity tests, then g3 (1e6 x 1 kbp semi-global) and 1e5 x 300 bp against the generic kernel, and the kernel
# trace of a g3 run.  Usage (through gpurun): bash scripts/wide_measure.sh [tag] [pmc]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-wide}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --timeout 240 -k "wide_kernel" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log | cut -c1-300
B="--cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0"
s() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); c=d['config']
print('$2', 'ms', round(d['ms_per_step'],2), 'fwd', round(c['main_kernel_ms'],2), 'allk', round(c['kernel_ms_per_step'],2), 'kind', d['roofline']['kernel'][:24], 'retried', c['retried_pairs'], 'ok', c['status_ok'])" || tail -3 $1.err; }
b() { n=$1; shift; timeout 400 python bench.py $B "$@" > $OUT/$n.json 2> $OUT/$n.json.err; s $OUT/$n.json $n; }
b g3_wide --config g3 --steps 3 --warmup 1
b g3_wide_w1 --config g3 --steps 3 --warmup 1 --opt wide_waves=1
b g3_wide_1phase --config g3 --steps 3 --warmup 1 --opt wide=3
b g3_generic --config g3 --steps 3 --warmup 1 --opt wide=0
b g300_wide --config g3 --steps 20 --warmup 2 --pairs 100000 --length 300
b g300_wide_w4 --config g3 --steps 20 --warmup 2 --pairs 100000 --length 300 --opt wide_waves=4
b g300_generic --config g3 --steps 20 --warmup 2 --pairs 100000 --length 300 --opt wide=0
b g3off_wide --config g3 --steps 3 --warmup 1 --pairs 20000 --no-adaptive
b g3off_generic --config g3 --steps 3 --warmup 1 --pairs 20000 --no-adaptive --opt wide=0
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/bench.py $B --config g3 --steps 2 --warmup 1 > $R/$OUT/stats.log 2>&1
if [ "${2:-}" = pmc ]; then
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/pmc_$N -- python3 $R/bench.py $B --config g3 --steps 1 --warmup 1 > $R/$OUT/pmc_$N.log 2>&1
done
fi
cd $R
grep -h "wfa_wide\|generic\|backtrace" $OUT/stats/*/*kernel_stats.csv | cut -c1-160
python3 - $OUT <<'PY'
import csv,glob,collections,sys
for f in glob.glob(sys.argv[1] + '/pmc_*/*/*counter_collection.csv'):
    acc=collections.defaultdict(float); cnt=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        kn=r['Kernel_Name']
        if 'wide' not in kn: continue
        key=(kn[kn.index('wfa_wide'):][:36], r['Counter_Name'])
        acc[key]+=float(r['Counter_Value']); cnt[key]+=1
    for k in sorted(acc): print(k, 'sum', acc[k], 'launches', cnt[k])
PY
