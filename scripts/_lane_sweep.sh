cd $GRAFT_REPO_ROOT
for n in 1000 4000 16000 50000 200000; do for o in lane=0 lane=2; do python3 bench.py --config c2 --pairs $n --steps 2000 --warmup 20 --cpu-sample 0 --host-entry 0 --latency 0 --cpu-all-cores 0 --opt $o 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('n=$n $o', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernels %.4f' % c['kernel_ms_per_step'], 'main %.4f' % c['main_kernel_ms'], d['roofline']['kernel'])
"; done; done
