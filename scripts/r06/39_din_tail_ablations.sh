#!/bin/bash
# Round 6: k_din_fused's tail ablations on config 3 (-DSPRK_DF_XP build, timing only): 128 no folded-row gathers, 256 no fc1, 512 no fc0 MFMAs, 896 all three
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_42}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
cp scripts/r06/libsparrow_hip_xp.so sparrowrecsys_amd/libsparrow_hip.so
for rep in 1 2; do
  for x in 0 128 256 512 896; do
    echo "SPRK_DF_XP=$x: $(SPRK_DF_XP=$x timeout 200 python bench.py --workload din_c3 --steps 60 --warmup 8 $STRICT 2>>$O/err.txt | tail -1 | python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('strict %.2f us | attention only %.2f us' % (r['avg_launch_us'], r['attention_only']['avg_launch_us']))")" | tee -a $O/timing.txt
  done
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
