#!/bin/bash
# What the driver runs at round end, on the current build: the GPU suite, smoke(), the bench command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_30
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -6 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
t0=$(date +%s); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | grep '^{"metric"' | tail -1 > $O/bench_driver_command.json; echo "bench wall $(( $(date +%s) - t0 )) s"
python -c "
import json
l=json.loads(open('$O/bench_driver_command.json').read())
print('driver: value %.4g one-batch %.4g frac %.4f hbm %.4f (%.2f us)' % (l['value'], l['value_one_batch_per_launch'], l['roofline']['frac'], l['roofline_hbm_resident']['frac'], l['roofline_hbm_resident']['avg_launch_us']))
for k,w in l['workloads'].items(): print(k, w['roofline']['kernel'][:30], '%.4g' % w['value'], '%.4f' % w['roofline']['frac'], w['roofline_mfma']['frac'])
print(l['roofline'].get('traffic'), l['roofline'].get('traffic_source'))"
