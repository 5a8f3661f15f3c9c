#!/bin/bash
# (The A/B libraries are not in the tree: build libsparrow_hip_mrold.so from commit 2a1e325.)
# Round 6: k_mlp_rows with the numerics' A operands as four coalesced loads per lane BEHIND the first gather (the tree) against round 5's sixteen
# 4-byte loads between the meeting and the gather (scripts/r06/libsparrow_hip_mrold.so = the build of commit 2a1e325); then the stamped timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_36}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stated_sizes.py -m gpu -x -q -k "mlp or wide or embedding or every_tile" 2>&1 | tail -3 | tee $O/pytest.txt
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('strict %.2f us = %.1f %% | oracle %s' % (r['avg_launch_us'], 100*r['frac'], l['config'].get('oracle_check_max_abs_err')))"; }
STRICT="--cpu-seconds 0 --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = new ]; then cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r06/libsparrow_hip_mrold.so sparrowrecsys_amd/libsparrow_hip.so; fi
    for w in widedeep_c5 embedding_mlp_ref; do
      echo "$v $w: $(timeout 300 python bench.py --workload $w --steps 200 --warmup 20 $STRICT 2>>$O/err.txt | tail -1 | get)" | tee -a $O/timing.txt
    done
  done
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
bash scripts/r06/34_mlp_rows_timeline.sh ${1:-r06_36} > $O/timeline_all.txt 2>&1
grep -A3 "first gather (us\|^bench\|per trip" $O/timeline_all.txt
