#!/bin/bash
# Round 6: k_mlp_rows' genre rows 544 bytes apart (the bank model's best stride, scripts/r06/lds_conflict_model.py) against the product's 528
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_21}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('strict %.2f us = %.1f %%' % (r['avg_launch_us'], 100*r['frac']))"; }
STRICT="--cpu-seconds 0 --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for rep in 1 2 3; do
  for v in rs136 product; do
    if [ $v = product ]; then cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r06/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so; fi
    for w in widedeep_c5 embedding_mlp_ref; do
      echo "$v $w: $(timeout 300 python bench.py --workload $w --steps 200 --warmup 20 $STRICT 2>>$O/err.txt | tail -1 | get)" | tee -a $O/timing.txt
    done
  done
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
