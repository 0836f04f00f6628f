#!/bin/bash
# Instruction counters of wfa_wide_kernel's two launches on g3: the default build against option wide_exact=1 (no packed path in the wide rows).
# Three passes over the 1e6 pairs per run (warm-up, step, census; the census pass takes the exact path in both).  Usage (through gpurun): bash scripts/wide_pmc2.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}
R=$(pwd); OUT=gpurun_out/widepmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="--cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 --config g3 --steps 1 --warmup 1"
for V in 0 1; do
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/$OUT/v$V -- python3 $R/bench.py $B --opt wide_exact=$V > $R/$OUT/v$V.log 2>&1
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for V in (0,1):
    for f in glob.glob(f'gpurun_out/widepmc/v{V}/*/*counter_collection.csv'):
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            kn=r['Kernel_Name']
            if 'wide' not in kn: continue
            acc[(kn[kn.index('wfa_wide'):][:36], r['Counter_Name'])]+=float(r['Counter_Value'])
        for k in sorted(acc): print('exact' if V else 'default', k, f"{acc[k]:.4g}")
PY
