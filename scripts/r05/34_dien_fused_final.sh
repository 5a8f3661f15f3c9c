#!/bin/bash
# Round 5, call 34: the final k_dien_fused (fence in front of every MFMA group, every lambda inlined): the DIEN / tail tests, the stress runs against
# the two-launch path (bit for bit), then DIEN.py's own shape strict and pipelined against SPRK_DIEN_FUSED=0, alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_34}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shape_sweep.py -m gpu -x -q -k "dien or tail" > $O/pytest_dien.log 2>&1; tail -3 $O/pytest_dien.log
for a in "16 7 65536" "10 5 65536" "16 20 20000" "16 7 4099" "10 50 777"; do timeout 300 python scripts/r05/dbg/dien_fused_stress.py $a ${RUNS:-120} 2>&1 | tail -1 | tee -a $O/stress.txt; done
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('%s | step %.3f us (sequence stage alone %.3f us) | value %.4g samples/s (%.3f us/step)' % (l['config'].get('kernel', r.get('kernel')), r.get('step_us_all_kernels', r['avg_launch_us']), r['avg_launch_us'], l['value'], l['ms_per_step']*1e3))"; }
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
PIPE="--cpu-seconds 0 --side-workloads= --no-hardware-probe --hbm-resident 0"
for rep in 1 2; do
  for sw in 1 0; do
    echo "strict    SPRK_DIEN_FUSED=$sw: $(SPRK_DIEN_FUSED=$sw timeout 300 python bench.py --workload dien_ref --steps 200 --warmup 20 $STRICT 2>$O/strict_$sw.err | tail -1 | get)" | tee -a $O/dien_ref.txt
  done
done
for sw in 1 0; do
  echo "pipelined SPRK_DIEN_FUSED=$sw: $(SPRK_DIEN_FUSED=$sw timeout 300 python bench.py --workload dien_ref --steps 200 --warmup 20 $PIPE 2>$O/pipe_$sw.err | tail -1 | get)" | tee -a $O/dien_ref.txt
done
