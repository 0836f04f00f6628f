#!/bin/bash
# One bench line of a configuration, summarised on one line.  Usage (GPU box): scripts/bench_line.sh <out dir> <config> [bench args...]
OUT=$1; C=$2; shift 2
mkdir -p $OUT
timeout 900 python bench.py --config $C --host-entry 0 --latency 0 --cpu-sample 0 "$@" > $OUT/bench_$C.json 2> $OUT/bench_$C.err
python3 -c "
import json; d=json.load(open('$OUT/bench_$C.json')); c=d['config']; r=d['roofline']
print('$C', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'allk', round(c['kernel_ms_per_step'],3), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'], 'ok', c['status_ok'], 'kernel', r['kernel'], 'frac', round(r['frac'],4), 'cells/pair', round(c['wf_cells_per_pair'],1))" || tail -3 $OUT/bench_$C.err
