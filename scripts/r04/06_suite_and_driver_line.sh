#!/bin/bash
# Round 4: the whole GPU suite on the current build, then the driver's own command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_06
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)|RuntimeWarning|eng = self|warnings.html" | tail -15 | tee $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench_driver_command.json
python - $O/bench_driver_command.json <<'PY'
import sys, json
l = json.loads(open(sys.argv[1]).read())
print({k: l.get(k) for k in ("metric", "value", "ms_per_step", "value_one_batch_per_launch")})
print('roofline', {k: l['roofline'].get(k) for k in ('kernel', 'frac', 'avg_launch_us')})
for k, w in l.get('workloads', {}).items():
    print(k, {x: w.get(x) for x in ('value', 'ms_per_step', 'value_one_batch_per_launch', 'kernel', 'oracle_check_max_abs_err')}, w['roofline'].get('avg_launch_us'), w['roofline'].get('step_us_all_kernels_strict'))
PY
