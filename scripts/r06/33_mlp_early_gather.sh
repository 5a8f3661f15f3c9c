#!/bin/bash
# (Not kept: needs profiles/r06/experiments/r06_33/k_mlp_rows_early_gather.patch applied; the =0 library is built with SPRK_BUILD_DEFINES=-DMR_EARLY_GATHER=0.)
# Round 6: k_mlp_rows with the first task's gather requested in FRONT of the image's staging (MR_EARLY_GATHER=1, the tree's build) against
# behind the workgroup's meeting (=0, rounds 2-5: scripts/r06/libsparrow_hip_mrlate.so built with SPRK_BUILD_DEFINES=-DMR_EARLY_GATHER=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_33}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stated_sizes.py -m gpu -x -q -k "mlp or wide or embedding or every_tile" 2>&1 | tail -3 | tee $O/pytest.txt
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('strict %.2f us = %.1f %% | oracle %s' % (r['avg_launch_us'], 100*r['frac'], l['config'].get('oracle_check_max_abs_err')))"; }
STRICT="--cpu-seconds 0 --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for rep in 1 2 3; do
  for v in early late; do
    if [ $v = early ]; then cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r06/libsparrow_hip_mrlate.so sparrowrecsys_amd/libsparrow_hip.so; fi
    for w in widedeep_c5 embedding_mlp_ref; do
      echo "$v $w: $(timeout 300 python bench.py --workload $w --steps 200 --warmup 20 $STRICT 2>>$O/err.txt | tail -1 | get)" | tee -a $O/timing.txt
    done
  done
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
