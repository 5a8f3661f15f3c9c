#!/bin/bash
# Round 4: k_din_fused variants on ONE box (boxes differ by ~8 %): SPRK_DF_OPT bit 1 = software-pipelined slot loop, bit 2 = the tail's
# embedding rows gathered in the prologue; against the two-launch path (SPRK_DIN_FUSED=0).  DIN tests under every variant first.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_05
mkdir -p $O
for opt in ${OPTS_TEST:-0 1 2 3}; do
  echo "=== DIN tests, SPRK_DF_OPT=$opt"; SPRK_DF_OPT=$opt timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_blocks.py tests/test_gpu_shape_sweep.py tests/test_gpu_host_api.py tests/test_gpu_stated_sizes.py -m gpu -q -k "din or DIN" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8
done
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    r = l['roofline']
    print('%-28s' % sys.argv[1].split('/')[-1], 'us/step %.2f' % (l['ms_per_step'] * 1e3), ' attention-only us %.2f' % r['avg_launch_us'],
          ' strict step us %.2f' % r.get('step_us_all_kernels', 0), ' err', l['config'].get('oracle_check_max_abs_err'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
STRICT="--cpu-seconds 0 --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
MBF="--cpu-seconds 0 --side-workloads= --no-hardware-probe"
for rep in 1 2; do
b c3_unfused_strict_$rep SPRK_DIN_FUSED=0 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT
for opt in ${OPTS:-0 1 2 3}; do b c3_opt${opt}_strict_$rep SPRK_DF_OPT=$opt python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT; done
done
b c3_unfused_mb SPRK_DIN_FUSED=0 python bench.py --workload din_c3 --steps 320 --warmup 32 $MBF
for opt in ${OPTS:-0 1 2 3}; do b c3_opt${opt}_mb SPRK_DIN_FUSED_MB=1 SPRK_DF_OPT=$opt python bench.py --workload din_c3 --steps 320 --warmup 32 $MBF; done
for opt in 0; do b c3_opt${opt}_mb1stream SPRK_DIN_FUSED_MB=1 SPRK_DF_OPT=$opt python bench.py --workload din_c3 --steps 320 --warmup 32 --overlap-streams 0 $MBF; done
b ref_unfused_strict SPRK_DIN_FUSED=0 python bench.py --workload din_ref --steps 120 --warmup 12 $STRICT
b ref_fused_strict SPRK_DIN_FUSED_ALWAYS=1 python bench.py --workload din_ref --steps 120 --warmup 12 $STRICT
b ref_default_strict python bench.py --workload din_ref --steps 120 --warmup 12 $STRICT
b ref_unfused_mb SPRK_DIN_FUSED=0 python bench.py --workload din_ref --steps 320 --warmup 32 $MBF
b ref_fused_mb SPRK_DIN_FUSED_MB=1 python bench.py --workload din_ref --steps 320 --warmup 32 $MBF
