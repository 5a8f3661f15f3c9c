#!/bin/bash
# Round 5, call 33: (a) is it the ORDER or the two wait states?  -DDNF_XP=65536: a bare sched_barrier in front of each MFMA group; 131072: an empty asm
# with a memory clobber (LDS reads may not cross; no wait state).  (b) the `s_nop 1` build (8192) over many launches at B = 65 536, both widths.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_33}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
for v in ${VARIANTS:-dnf65536 dnf131072}; do
  cp scripts/r05/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so || continue
  echo "== $v" | tee -a $O/stress.txt
  timeout 200 python scripts/r05/dbg/dien_fused_stress.py 16 7 4099 40 2>&1 | tail -1 | tee -a $O/stress.txt
done
for v in ${STRESS:-dnf8192}; do
  cp scripts/r05/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so || continue
  echo "== $v (stress)" | tee -a $O/stress.txt
  timeout 300 python scripts/r05/dbg/dien_fused_stress.py 16 7 65536 ${RUNS:-150} 2>&1 | tail -1 | tee -a $O/stress.txt
  timeout 300 python scripts/r05/dbg/dien_fused_stress.py 10 5 65536 ${RUNS:-150} 2>&1 | tail -1 | tee -a $O/stress.txt
  timeout 300 python scripts/r05/dbg/dien_fused_stress.py 16 20 20000 ${RUNS:-150} 2>&1 | tail -1 | tee -a $O/stress.txt
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
