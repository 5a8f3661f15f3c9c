#!/bin/bash
# Round 6: what k_mlp_rows' lane map costs the texture path -- MR_XP=128 (timing only, wrong results): the big rows requested with a QUAD of consecutive
# lanes on one row's 64 consecutive bytes (same lines per instruction as the product, whose quad touches four different rows); MR_XP=4: not loaded at all
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_40}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('strict %.2f us' % (r['avg_launch_us']))"; }
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for rep in 1 2; do
  for v in product mrq mrnoload; do
    if [ $v = product ]; then cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r06/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so; fi
    for w in widedeep_c5 embedding_mlp_ref; do
      echo "$v $w: $(timeout 300 python bench.py --workload $w --steps 200 --warmup 20 $STRICT 2>>$O/err.txt | tail -1 | get)" | tee -a $O/timing.txt
    done
  done
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
