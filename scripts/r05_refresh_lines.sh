#!/bin/bash
# after the closing run has committed its PMC profiles: smoke, the headline line with its eleven other legs, c5s and p242 -- so that the stored lines name profiles of the same sources
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_n; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; python3 -c "
import json; d=json.loads([l for l in open('$OUT/bench_c3.json') if l.startswith('{')][-1]); r=d['roofline']; c=d['config']
print('c3', round(d['value']/1e6,2), round(d['ms_per_step'],3), r['kernel'], r['traffic_source'], r['traffic_stale'], round(r['frac'],4), r.get('frac_on_traffic'), r['secondary']['source'], r['secondary']['stale'], c.get('clock_mhz_under_load'))
print({k:(round(v.get('value',0),2), v.get('status_ok')) for k,v in c['other_configs'].items()})"
for c in c5s p242; do timeout 900 python bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err; python3 -c "
import json; d=json.loads([l for l in open('$OUT/bench_$c.json') if l.startswith('{')][-1]); r=d['roofline']; print('$c', round(d['value'],2), round(d['ms_per_step'],3), r['kernel'], r['traffic_source'], r['traffic_stale'], round(r['frac'],4), r.get('traffic'))"; done
