#!/bin/bash
# Round 6, call 7: the product WITHOUT the s_nop fence, the un-scale scalars through lone_scalar() (dyn_split.h): no packed-f32 instruction takes the
# high dword of a VGPR src1 any more (scripts/isa/isa_pk_opsel.py).  DIEN / tail tests; both DIEN paths over many launches at three shapes bit for bit
# + the fp64 oracle on every 8th tile; the timing of DIEN.py's shape against round 5's fenced build (37.5 / 24.4 us).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_10}
mkdir -p $O
python scripts/isa/isa_pk_opsel.py --so sparrowrecsys_amd/libsparrow_hip.so | tail -1 | tee $O/isa_scan.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shape_sweep.py -m gpu -x -q -k "dien or tail" > $O/pytest_dien.log 2>&1; tail -2 $O/pytest_dien.log
for a in "16 7 65536" "10 5 65536" "16 20 20000"; do timeout 400 python scripts/r05/dbg/dien_fused_stress.py $a ${RUNS:-100} 2>&1 | tail -3 | tee -a $O/stress.txt; done
get() { python -c "import sys,json;l=json.loads(sys.stdin.read());r=l['roofline'];print('%s | step %.3f us (sequence stage alone %.3f us) | value %.4g samples/s (%.3f us/step)' % (l['config'].get('kernel', r.get('kernel')), r.get('step_us_all_kernels', r['avg_launch_us']), r['avg_launch_us'], l['value'], l['ms_per_step']*1e3))"; }
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for rep in 1 2; do
  for sw in 1 0; do
    echo "strict    SPRK_DIEN_FUSED=$sw: $(SPRK_DIEN_FUSED=$sw timeout 300 python bench.py --workload dien_ref --steps 200 --warmup 20 $STRICT 2>$O/strict_$sw.err | tail -1 | get)" | tee -a $O/dien_ref.txt
  done
done
