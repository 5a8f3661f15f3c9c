#!/bin/bash
# Round 4: k_din_fused with the two-round-trip prologue (ids by LDS-DMA; every second-round-trip load hidden; the tail's image behind
# them, landing under the first trip).  DIN tests, strict and several-batches numbers, then the stamped timeline with the -DSPRK_DF_XP
# build of the same tree (scripts/r04/libsparrow_hip_xp.so, built beside the product library).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_16
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q -k "din or DIN" > $O/pytest_din.log 2>&1
tail -3 $O/pytest_din.log
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%s: step %.2f us   dominant kernel %.2f us   value %.4g' % (sys.argv[2], l['roofline']['step_us_all_kernels'], l['roofline']['avg_launch_us'], l['value']))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
for i in 1 2; do
  timeout 200 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT 2>$O/strict$i.err | tail -1 > $O/strict$i.json
  show $O/strict$i.json "strict"
done
SPRK_DIN_FUSED=0 timeout 200 python bench.py --workload din_c3 --steps 120 --warmup 12 $STRICT 2>$O/strict_two.err | tail -1 > $O/strict_two.json
show $O/strict_two.json "strict, two launches (SPRK_DIN_FUSED=0)"
MBF="--cpu-seconds 0 --no-check --hbm-resident 0 --side-workloads= --no-hardware-probe"
for mb in 0 1; do
  SPRK_DIN_FUSED_MB=$mb timeout 200 python bench.py --workload din_c3 --steps 128 --warmup 16 $MBF 2>$O/mb$mb.err | tail -1 > $O/mb$mb.json
  show $O/mb$mb.json "16 batches per launch, FUSED_MB=$mb"
done
SPRK_DIN_MB_ATTN_FUSED=1 timeout 200 python bench.py --workload din_c3 --steps 128 --warmup 16 $MBF 2>$O/mb_attn_fused.err | tail -1 > $O/mb_attn_fused.json
show $O/mb_attn_fused.json "16 batches per launch, pipeline with k_din_fused as the attention launch"
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
cp scripts/r04/libsparrow_hip_xp.so sparrowrecsys_amd/libsparrow_hip.so
SPRK_DF_XP=1024 SPRK_DF_TS_FILE=$O/ts.bin timeout 200 python bench.py --workload din_c3 --steps 40 --warmup 8 $STRICT 2>$O/ts.err | tail -1 > $O/ts.json
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
python scripts/r04/din_fused_timeline.py $O/ts.bin $O/ts.json | tee $O/timeline.txt
