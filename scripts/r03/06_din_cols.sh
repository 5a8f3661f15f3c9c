#!/bin/bash
# Round 3, GPU call 6: k_din_attn_cols (16 samples per MFMA tile, static weight operand) -- DIN tests, then A/B on config 3 and
# on DIN.py's literal shape.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_06
mkdir -p $O
echo "=== DIN tests"; timeout 900 python -m pytest tests -m gpu -q -x -k "din or DIN" 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -15 | tee $O/pytest_din.log
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    r = l['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.4g' % l['value'], 'us/step %.2f' % (l['ms_per_step'] * 1e3), 'attention us %.2f frac %.3f' % (r['avg_launch_us'], r['frac']),
          'strict step us %.2f' % r['step_us_all_kernels'], 'err', l['config']['oracle_check_max_abs_err'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
b c3_cols python bench.py --workload din_c3 --cpu-seconds 0 --steps 320 --warmup 32
b c3_wave SPRK_DIN_COLS=0 python bench.py --workload din_c3 --cpu-seconds 0 --steps 320 --warmup 32
b ref_cols python bench.py --workload din_ref --cpu-seconds 0 --steps 320 --warmup 32
b ref_wave SPRK_DIN_COLS=0 python bench.py --workload din_ref --cpu-seconds 0 --steps 320 --warmup 32
tail -3 $O/c3_cols.err
