#!/bin/bash
# Round 6: k_din_fused with wave priorities alternating between the two waves of a SIMD (DF_PRIO = 1, 2, 3) against the product: config 3 strict
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_23}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('strict %.2f us = %.1f %% | attention only %.2f us | oracle %s' % (r['avg_launch_us'], 100*r['frac'], r['attention_only']['avg_launch_us'], l['config'].get('oracle_check_max_abs_err')))"; }
STRICT="--cpu-seconds 0 --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for rep in 1 2 3; do
  for v in ${VARIANTS:-prio1 prio2 prio3 product}; do
    if [ $v = product ]; then cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r06/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so; fi
    echo "$v: $(timeout 300 python bench.py --workload din_c3 --steps 100 --warmup 10 $STRICT 2>>$O/err.txt | tail -1 | get)" | tee -a $O/timing.txt
  done
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
