#!/bin/bash
# Round 4: dyn_split8 as v_pk_mul + v_cvt_pk_f16_f32 + v_fma_mix_f32 (11.9 cycles per value) instead of v_fma_mixlo/hi_f16 (17.4): the GPU
# suite (same bits expected: the twin-kernel equalities and every golden), then the kernels that split activations per sample.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_10
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for w in deepfm_c2 widedeep_c5 dien_ref din_ref embedding_mlp_ref deepfm_c4; do
  timeout 300 python bench.py --workload $w --steps 200 --warmup 20 $STRICT 2>/dev/null | tail -1 > $O/$w.json
  python - $O/$w.json $w <<'PY'
import sys, json
l = json.loads(open(sys.argv[1]).read()); r = l['roofline']
print('%-20s strict us %.2f  step us %s  frac %.3f' % (sys.argv[2], r['avg_launch_us'], r.get('step_us_all_kernels'), r['frac']))
PY
done
