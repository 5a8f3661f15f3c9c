#!/bin/bash
# Round 4: does the persistent k_din_fused<MB> run at the fabric's rate for ALL its bytes?  The same launch without the folded rows'
# gathers (33.5 MB per batch out of L2; -DSPRK_DF_XP build, SPRK_DF_XP=128: garbage scores, the time is the point).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_19}
mkdir -p $O
show() { python - $1 "$2" <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%s: value %.4g   (%.2f us per batch)' % (sys.argv[2], l['value'], 32768 / l['value'] * 1e6))
except Exception as e:
    print('%s FAILED %s' % (sys.argv[2], e))
PY
}
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "din or DIN" > $O/pytest_din.log 2>&1
tail -1 $O/pytest_din.log
MBF="--cpu-seconds 0 --no-check --hbm-resident 0 --side-workloads= --no-hardware-probe"
for mb in 0 1 0 1; do
  SPRK_DIN_FUSED_MB=$mb timeout 200 python bench.py --workload din_c3 --steps 128 --warmup 16 $MBF 2>$O/mb$mb.err | tail -1 > $O/mb$mb.json
  show $O/mb$mb.json "16 batches per launch, FUSED_MB=$mb"
done
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/libsparrow_hip_product.so
cp scripts/r04/libsparrow_hip_xp.so sparrowrecsys_amd/libsparrow_hip.so
for xp in 0 128; do
  SPRK_DF_XP=$xp SPRK_DIN_FUSED_MB=1 timeout 200 python bench.py --workload din_c3 --steps 128 --warmup 16 $MBF 2>$O/xp$xp.err | tail -1 > $O/xp$xp.json
  show $O/xp$xp.json "XP build, FUSED_MB=1, SPRK_DF_XP=$xp"
done
cp /tmp/libsparrow_hip_product.so sparrowrecsys_amd/libsparrow_hip.so
