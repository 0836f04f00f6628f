cd $GRAFT_REPO_ROOT
timeout 800 python scripts/lane_check.py > gpurun_out/lane_check.log 2>&1
grep -c " ok" gpurun_out/lane_check.log; grep -i "fail\|error\|Traceback" gpurun_out/lane_check.log | head -5; tail -4 gpurun_out/lane_check.log
for o in ${LANE_OPTS:-lane=2}; do python3 bench.py --config c2 --cpu-sample 0 --host-entry 0 --latency 0 --opt $o 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$o', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernels %.4f' % c['kernel_ms_per_step'], 'main %.4f' % c['main_kernel_ms'], d['roofline']['kernel'])
"; done
