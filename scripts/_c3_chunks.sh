cd $GRAFT_REPO_ROOT
for o in "" "--opt chunk_pairs=900000" "--opt chunk_pairs=800000" "--opt chunk_pairs=950000"; do python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --host-entry 0 --latency 0 --cpu-all-cores 0 $o 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('[$o]', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernels %.4f' % c['kernel_ms_per_step'], 'main %.4f' % c['main_kernel_ms'], 'launches', c['launches_per_step'], 'retried', c['retried_pairs'], d['roofline']['kernel'])
"; done
