#!/bin/bash
# Round 4: the 800-candidate REST request after the batcher-thread fix and the handler's own header parser: 1, 4, 8, 16 clients.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_24}
mkdir -p $O
for c in 1 1 4 8 16; do
  echo "clients=$c: $(timeout 200 python scripts/bench_serving.py --clients $c --seconds 2 | tee $O/serving_clients$c.json | cut -c1-200)"
done
