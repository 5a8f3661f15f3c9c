#!/bin/bash
# Round 6: configs 4 / 5 with a working set of 4 x the Infinity Cache (bench.py hbm_cycle): strict and 16 batches per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_12}
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
for w in deepfm_c4 deepfm_v2_c4 widedeep_c5; do
  for nb in 0 8; do
    timeout 600 python bench.py --workload $w --steps 200 --warmup 20 --input-batches $nb $STRICT 2>$O/$w.err | tail -1 > $O/${w}_nb$nb.json
    python - $O/${w}_nb$nb.json <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read()); r=l["roofline"]
print("%-14s batches cycled %3d | working set %7.0f MB | strict %.2f us = %.1f %% | hbm-side %.0f GB/s | value %.3g samples/s" % (l["config"]["workload"].split(":")[0], l["config"]["input_batches_cycled"], r.get("working_set_mb", 0), r["avg_launch_us"], 100*r["frac"], r.get("hbm_side_GBps", 0), l["value"]))
PY
  done
done 2>&1 | tee $O/summary.txt
