#!/bin/bash
# Round 6, call 4: the failing build with a dump behind block 7's consumers -- which value is the first to be wrong in a bad tile?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_05}
mkdir -p $O
cp sparrowrecsys_amd/libsparrow_hip.so /tmp/product.so
cp scripts/r06/libsparrow_hip_${V:-dump7}.so sparrowrecsys_amd/libsparrow_hip.so
timeout 600 python scripts/r06/dien_dump_run.py ${RUNS:-9} 2>&1 | tail -40 | tee $O/dump.txt
cp /tmp/product.so sparrowrecsys_amd/libsparrow_hip.so
