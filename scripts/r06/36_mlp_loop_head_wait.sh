#!/bin/bash
# (Not kept; mrmid = the tree, new = the tree with `__builtin_amdgcn_s_waitcnt(0x0F70)` in front of k_mlp_rows' task loop / the small-column block moved in front of the gather: DESIGN 7.4.)
# Round 6: k_mlp_rows with everything of the prologue waited for once in front of the task loop (the tree) against the build whose loop waited
# vmcnt(19) / vmcnt(16) in front of every trip's numerics MFMAs (libsparrow_hip_mrmid.so) and against round 5's (libsparrow_hip_mrold.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_37}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stated_sizes.py -m gpu -x -q -k "mlp or wide or embedding or every_tile" 2>&1 | tail -3 | tee $O/pytest.txt
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('strict %.2f us = %.1f %% | oracle %s' % (r['avg_launch_us'], 100*r['frac'], l['config'].get('oracle_check_max_abs_err')))"; }
STRICT="--cpu-seconds 0 --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for rep in 1 2 3; do
  for v in new mid old; do
    if [ $v = new ]; then cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r06/libsparrow_hip_mr$v.so sparrowrecsys_amd/libsparrow_hip.so; fi
    for w in widedeep_c5 embedding_mlp_ref; do
      echo "$v $w: $(timeout 300 python bench.py --workload $w --steps 200 --warmup 20 $STRICT 2>>$O/err.txt | tail -1 | get)" | tee -a $O/timing.txt
    done
  done
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
