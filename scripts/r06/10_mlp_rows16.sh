#!/bin/bash
# Round 6: k_mlp_rows16 (12 or 16 waves per workgroup, one task in flight per wave) against the eight-wave k_mlp_rows: parity tests, then config 5
# and EmbeddingMLP.py's shape, strict, alternating order.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_13}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stated_sizes.py tests/test_gpu_shape_sweep.py -m gpu -x -q -k "mlp or wide or widedeep or embedding" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('%s | strict %.2f us = %.1f %% | value %.4g samples/s' % (l['config'].get('kernel'), r['avg_launch_us'], 100*r['frac'], l['value']))"; }
STRICT="--cpu-seconds 0 --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for rep in 1 2; do
  for v in w12 w16 w8; do
    if [ $v = w16 ]; then cp scripts/r06/libsparrow_hip_w16.so sparrowrecsys_amd/libsparrow_hip.so; else cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; fi
    export SPRK_MLP_ROWS16=1; [ $v = w8 ] && export SPRK_MLP_ROWS16=0
    for w in widedeep_c5 embedding_mlp_ref; do
      echo "$v $w: $(timeout 300 python bench.py --workload $w --steps 200 --warmup 20 $STRICT 2>>$O/err.txt | tail -1 | get)" | tee -a $O/timing.txt
    done
  done
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
