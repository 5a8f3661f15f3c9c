#!/bin/bash
# Round 4: the 800-candidate REST request (RecForYouProcess.java:34,113-138) end to end on the GPU box: latency from 1 and 8 clients,
# with the round-4 changes (inline forward when idle, a batching wait only while another request is arriving) switched off and on.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_09
mkdir -p $O
for cfg in "1 1" "0 0" "0 1" "1 0"; do
  set -- $cfg
  for c in 1 8; do
    echo "inline=$1 adaptive_wait=$2 clients=$c: $(SPRK_SERVING_INLINE=$1 SPRK_SERVING_ADAPTIVE_WAIT=$2 timeout 200 python scripts/bench_serving.py --clients $c --seconds 2 | tee $O/serving_inline$1_adaptive$2_clients$c.json | cut -c1-175)"
  done
done
