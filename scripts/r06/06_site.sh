#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_06}
mkdir -p $O
timeout 600 scripts/ubench/dien_site_repro 20000 > $O/dien_site_repro.txt 2>&1; cat $O/dien_site_repro.txt | cut -c1-230
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
for v in ${VARIANTS:-rA pV}; do
  cp scripts/r06/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so || continue
  timeout 300 python scripts/r06/dien_seq_stress.py 16 7 65536 ${RUNS:-40} $v 2>&1 | tail -1 | tee -a $O/stress.txt
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
