#!/bin/bash
# Round 4, GPU call 3: k_din_fused, first correctness run (DIN tests through every attention / tail path) and a first timing.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_03
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "din or DIN" 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -25
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    r = l['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.4g' % l['value'], 'us/step %.2f' % (l['ms_per_step'] * 1e3), 'kernel us %.2f frac %.3f' % (r['avg_launch_us'], r['frac']),
          'strict step us %.2f' % r.get('step_us_all_kernels', 0), 'err', l['config'].get('oracle_check_max_abs_err'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
STRICT="--cpu-seconds 0 --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
b c3_fused_strict python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT
b c3_unfused_strict SPRK_DIN_FUSED=0 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT
b c3_fused_mb python bench.py --workload din_c3 --steps 320 --warmup 32 --cpu-seconds 0 --side-workloads= --no-hardware-probe
b c3_unfused_mb SPRK_DIN_FUSED=0 python bench.py --workload din_c3 --steps 320 --warmup 32 --cpu-seconds 0 --side-workloads= --no-hardware-probe
b ref_fused_strict python bench.py --workload din_ref --steps 120 --warmup 12 $STRICT
b ref_unfused_strict SPRK_DIN_FUSED=0 python bench.py --workload din_ref --steps 120 --warmup 12 $STRICT
