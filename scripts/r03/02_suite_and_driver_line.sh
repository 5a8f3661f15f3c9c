#!/bin/bash
# Round 3, GPU call 2: the whole GPU suite on the current build, smoke, and the driver's bench command with its wall time.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_02
mkdir -p $O
echo "=== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -15 | tee $O/pytest_gpu.log
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "=== bench default"; t0=$(date +%s.%N); timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc) s rc=$?"; tail -3 $O/bench_driver.err
tail -1 $O/bench_driver.json | python -c "
import sys, json
l = json.loads(sys.stdin.readline())
print('value', l['value'], 'one-batch', l['value_one_batch_per_launch'], 'frac', l['roofline']['frac'])
print('hbm_resident', {k: l['roofline_hbm_resident'][k] for k in ('frac','avg_launch_us','working_set_mb','frac_16_batches_per_launch')})
for k, w in l.get('workloads', {}).items():
    print(k, w['kernel'], w['value'], w['ms_per_step'], w['roofline']['frac'], w['roofline']['avg_launch_us'], w['oracle_check_max_abs_err'])
print('cpu', l['cpu_baseline']['value'], l['cpu_baseline']['cores'])
"
