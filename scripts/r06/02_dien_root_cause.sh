#!/bin/bash
# Round 6, call 2: (i) a load landing in SrcA / SrcB of the LAST MFMA of a dependent chain (scripts/ubench/mfma_chain_srcab_war.hip);
# (ii) more wait states at the two classes that helped in call 1 (raw, war), and ONE site (the fragment load behind block 7's chain).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_02}
mkdir -p $O
timeout 600 scripts/ubench/mfma_chain_srcab_war 20000 > $O/mfma_chain_srcab_war.txt 2>&1; cat $O/mfma_chain_srcab_war.txt | cut -c1-200
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
for v in ${VARIANTS:-none war32 war64 raw32 e1_64 rawwar}; do
  cp scripts/r06/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so || continue
  timeout 300 python scripts/r06/dien_seq_stress.py 16 7 65536 ${RUNS:-40} $v 2>&1 | tail -1 | tee -a $O/stress.txt
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
