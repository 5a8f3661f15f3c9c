#!/bin/bash
# everything the driver runs at round end, on the current tree: smoke(), the GPU test suite, the default bench command
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_final
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "NCCL\|RCCL" | tail -3
s=$(date +%s)
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "NCCL\|RCCL" > gpurun_out/r03_final/pytest_gpu.log
tail -4 gpurun_out/r03_final/pytest_gpu.log
echo "pytest seconds: $(( $(date +%s) - s ))"
s=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_final/bench_driver.json 2> gpurun_out/r03_final/bench_driver.err
echo "bench seconds: $(( $(date +%s) - s ))"
python - <<'PY'
import json
l=json.loads([x for x in open('gpurun_out/r03_final/bench_driver.json').read().splitlines() if x.startswith('{"metric"')][-1])
print('value %.4g frac %.4f hbm %.4f' % (l['value'], l['roofline']['frac'], l['roofline_hbm_resident']['frac']))
print('hardware', l.get('hardware'))
print('cpu', l['cpu_baseline']['value'], l['cpu_baseline'].get('cpu_model'))
for k,v in l['workloads'].items(): print(k, '%.4g'%v['value'], '%.4f'%v['roofline']['frac'])
PY
