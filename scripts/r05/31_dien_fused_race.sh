#!/bin/bash
# Round 5, call 31: k_dien_fused<16,...> gave run-to-run different scores for WHOLE tiles (one wave each); D = 10 never.  Which ingredient?
# Experiment builds (-DDNF_XP=bits, scripts/r05/build_variant.sh): 1 tail gathers behind the last step, 2 full wait + nops in front of the tail,
# 4 run-time D mask, 8 eight waves per workgroup (256 VGPRs, no spills).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_31}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
for v in ${SKIP_PRODUCT:+} $( [ -n "${SKIP_PRODUCT:-}" ] || echo product ) ${VARIANTS:-dnf1 dnf2 dnf4 dnf8}; do
  if [ $v = product ]; then cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r05/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so || { echo "no library for $v"; continue; }; fi
  echo "== $v" | tee -a $O/race.txt
  timeout 200 python scripts/r05/dbg/dien_fused_diff.py 16 7 4099 2>&1 | grep "^run\|rerun" | cut -c1-120 | tee -a $O/race.txt
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
