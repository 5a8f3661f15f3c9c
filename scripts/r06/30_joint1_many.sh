#!/bin/bash
# (Not kept: needs profiles/r06/experiments/r06_30/joint1_persistent_many.patch applied to sparrowrecsys_amd/csrc and a rebuild.)
# Round 6: several batches per launch on the persistent k_deepfm_v2_joint1<MB> (SPRK_V2J1_MANY=1, the default of this build) against
# k_deepfm_v2_joint_many (SPRK_V2J1_MANY=0): parity (forward_many == launch per batch, bit for bit), then the driver's line both ways.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_30}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q -k "many or predict" 2>&1 | tail -5 | tee $O/pytest_many.txt
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('value %.3f G/s | %.3f us/step | strict %.2f us | hbm-resident 16 batches: %s us' % (l['value']/1e9, 1e3*l['ms_per_step'], r['avg_launch_us'], r.get('hbm_resident_us_16_batches')))"; }
for rep in 1 2 3; do
  for v in 1 0; do
    echo "SPRK_V2J1_MANY=$v: $(SPRK_V2J1_MANY=$v timeout 600 python bench.py --steps 400 --warmup 40 --cpu-seconds 0 --side-workloads= --no-hardware-probe --variants 0 2>>$O/err.txt | tail -1 | tee -a $O/lines_$v.jsonl | get)" | tee -a $O/timing.txt
  done
done
