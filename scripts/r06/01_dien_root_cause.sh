#!/bin/bash
# Round 6, call 1: the DIEN flaky tiles (docs/open_issue_dien_tiles.md), VERDICT r05 item 1.  (i) the MFMA -> VALU RAW microbenchmark ADVICE r05
# asked for; (ii) the UNFENCED build with its schedule kept and wait states inserted at ONE class of places in the assembly
# (scripts/r06/isa_patch_build.py): none (control) / raw / war / pre / mid / wait0 -- which class turns it clean?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_01}
mkdir -p $O
timeout 600 scripts/ubench/mfma_to_valu_raw 2000 > $O/mfma_to_valu_raw.txt 2>&1; grep -c "wrong" $O/mfma_to_valu_raw.txt; grep -v " 0 wrong" $O/mfma_to_valu_raw.txt | head -40
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
for v in ${VARIANTS:-none raw war pre mid wait0 product}; do
  if [ $v = product ]; then cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so; else cp scripts/r06/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so || continue; fi
  timeout 300 python scripts/r06/dien_seq_stress.py 16 7 65536 ${RUNS:-40} $v 2>&1 | tail -1 | tee -a $O/stress.txt
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
