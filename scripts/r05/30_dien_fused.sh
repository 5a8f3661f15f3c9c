#!/bin/bash
# Round 5, call 30: DIEN in ONE launch (k_dien_fused.h: the recurrence, then k_din_tail's register chain as the same wave's epilogue) against the two
# launches (SPRK_DIEN_FUSED=0: k_dien_seq_mfma -> final states in HBM -> k_din_tail): the DIEN / DIN-tail parity tests, then DIEN.py's own shape
# (hist_len 5, emb_dim 10, B = 65 536) strict (one batch per launch, HIP events) and pipelined, alternating order.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_30}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shape_sweep.py -m gpu -x -q -k "dien or tail" > $O/pytest_dien.log 2>&1; tail -3 $O/pytest_dien.log
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('%s | step %.3f us (sequence stage alone %.3f us) | value %.4g samples/s (%.3f us/step)' % (l['config'].get('kernel', r.get('kernel')), r.get('step_us_all_kernels', r['avg_launch_us']), r['avg_launch_us'], l['value'], l['ms_per_step']*1e3))"; }
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
PIPE="--cpu-seconds 0 --side-workloads= --no-hardware-probe --hbm-resident 0"
for rep in 1 2 3; do
  for sw in 1 0; do
    echo "strict    SPRK_DIEN_FUSED=$sw: $(SPRK_DIEN_FUSED=$sw timeout 300 python bench.py --workload dien_ref --steps 200 --warmup 20 $STRICT 2>$O/strict_$sw.err | tail -1 | get)" | tee -a $O/dien_ref.txt
  done
done
for sw in 1 0; do
  echo "pipelined SPRK_DIEN_FUSED=$sw: $(SPRK_DIEN_FUSED=$sw timeout 300 python bench.py --workload dien_ref --steps 200 --warmup 20 $PIPE 2>$O/pipe_$sw.err | tail -1 | get)" | tee -a $O/dien_ref.txt
done
