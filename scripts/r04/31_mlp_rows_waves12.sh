#!/bin/bash
# Round 4: k_mlp_rows with twelve waves per workgroup (3 per SIMD, <= 168 VGPRs, a task requests its own rows and waits: the other waves
# hide the round trip) against eight (2 per SIMD, the next task's rows prefetched into 64 registers): SPRK_MLP_WAVES.
# RESULT (profiles/r04/experiments/r04_31): the same time (config 5 41.6-41.8 against 41.3-41.6 us, EmbeddingMLP 24.5-24.6 both) -- neither
# the prefetch nor a third wave per SIMD matters, the kernel is bound by the issue of its own instructions (937 VALU + 112 MFMA + 128
# ds_read_b128 per 16 samples: ~2 100 of the 2 700 cycles a CU spends per task).  The twelve-wave form and its switch are NOT in the
# tree (they doubled the kernel's instantiations); this script documents the experiment.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_31}
mkdir -p $O
SPRK_MLP_WAVES=12 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stated_sizes.py -m gpu -x -q -k "mlp or wide or Wide or embedding" > $O/pytest_mlp12.log 2>&1
tail -1 $O/pytest_mlp12.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%-30s kernel %.2f us   frac %.3f   value %.4g' % (sys.argv[2], l['roofline']['avg_launch_us'], l['roofline']['frac'], l['value']))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
for wv in 12 8 12 8; do
  for w in widedeep_c5 embedding_mlp_ref; do
    SPRK_MLP_WAVES=$wv timeout 200 python bench.py --workload $w --steps 100 --warmup 10 $STRICT 2>$O/${w}_w$wv.err | tail -1 > $O/${w}_w$wv.json
    show $O/${w}_w$wv.json "$w waves=$wv"
  done
done
