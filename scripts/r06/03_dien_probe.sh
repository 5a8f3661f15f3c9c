#!/bin/bash
# Round 6, call 3: a probe in the failing build's assembly at the consumer of block 7's result: are the bias / un pair / chain result the kernel is
# about to use what a re-read / recomputation gives?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_03}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
for v in ${VARIANTS:-probe7}; do
  cp scripts/r06/libsparrow_hip_$v.so sparrowrecsys_amd/libsparrow_hip.so || continue
  timeout 300 python scripts/r06/dien_probe_run.py 16 7 65536 ${RUNS:-30} $v 2>&1 | tail -2 | tee -a $O/probe.txt
done
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
