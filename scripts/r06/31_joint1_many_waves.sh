#!/bin/bash
# (Not kept: needs profiles/r06/experiments/r06_30/joint1_persistent_many.patch applied to sparrowrecsys_amd/csrc and a rebuild.)
# Round 6: the persistent k_deepfm_v2_joint1<MB> with 12 / 8 waves per CU and its weight fragments held in registers across tasks
# (V2J1_WAVES_MB, V2J1_MB_KEEP) against sixteen waves re-reading them per task (the build in the tree) and k_deepfm_v2_joint_many
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_31}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('value %.3f G/s | %.3f us/step | hbm-resident 16 batches: %.2f us' % (l['value']/1e9, 1e3*l['ms_per_step'], r.get('hbm_resident_us_16_batches')))"; }
for v in mb12k1 mb12k0 mb8k1; do
  cp scripts/r06/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so
  echo "$v: $(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k 'several_batches_per_launch and not pairs and not din' 2>&1 | tail -1)" | tee -a $O/timing.txt
done
for rep in 1 2; do
  for v in mb12k1 mb12k0 mb8k1 product old; do
    if [ $v = product -o $v = old ]; then cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r06/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so; fi
    e=1; [ $v = old ] && e=0
    echo "$v: $(SPRK_V2J1_MANY=$e timeout 600 python bench.py --steps 400 --warmup 40 --cpu-seconds 0 --side-workloads= --no-hardware-probe --variants 0 2>>$O/err.txt | tail -1 | tee -a $O/lines_$v.jsonl | get)" | tee -a $O/timing.txt
  done
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
