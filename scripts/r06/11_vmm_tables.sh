#!/bin/bash
# Round 6: derived gather tables >= 256 MB as one hipMemCreate allocation (host_engine.h table_alloc) against hipMalloc: strict launches, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_16}
mkdir -p $O
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('%s | strict %.2f us = %.1f %% | value %.4g samples/s' % (l['config'].get('kernel'), r['avg_launch_us'], 100*r['frac'], l['value']))"; }
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for rep in 1 2 3; do
  for vmm in 1 0; do
    export SPRK_VMM_TABLES=$vmm
    echo "vmm=$vmm c2_hbm:      $(timeout 300 python bench.py --steps 400 --warmup 40 --big-vocab 8388608 --input-batches 32 $STRICT 2>>$O/err.txt | tail -1 | get)" | tee -a $O/timing.txt
    if [ $rep = 1 ]; then
      echo "vmm=$vmm deepfm_v2_c4: $(timeout 300 python bench.py --workload deepfm_v2_c4 --steps 200 --warmup 20 $STRICT 2>>$O/err.txt | tail -1 | get)" | tee -a $O/timing.txt
      echo "vmm=$vmm deepfm_c4:    $(timeout 300 python bench.py --workload deepfm_c4 --steps 200 --warmup 20 $STRICT 2>>$O/err.txt | tail -1 | get)" | tee -a $O/timing.txt
    fi
  done
done
unset SPRK_VMM_TABLES
timeout 900 python -m pytest tests/test_gpu_stated_sizes.py tests/test_gpu_host_api.py -m gpu -x -q -k "config4 or config2 or host or destroy or lifetime" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
