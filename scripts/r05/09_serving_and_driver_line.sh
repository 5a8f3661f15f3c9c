#!/bin/bash
# Round 5, call 9: (a) the REST shim behind N front processes + ONE engine process (serving.serve_workers) under 8 clients, and one client;
# (b) the driver's command on this tree: wall time, the line's size, and the scalars that now live in the keys the driver keeps.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05_09}
mkdir -p $O
for w in 1 8 12 16; do
  timeout 300 python scripts/bench_serving.py --clients 8 --seconds 4 --workers $w > $O/serving_w${w}_c8.json 2>$O/serving_w$w.err; cut -c1-230 $O/serving_w${w}_c8.json
done
timeout 200 python scripts/bench_serving.py --clients 1 --seconds 3 --workers 12 > $O/serving_w12_c1.json 2>/dev/null; cut -c1-230 $O/serving_w12_c1.json
timeout 200 python scripts/bench_serving.py --clients 1 --seconds 3 --workers 1 > $O/serving_w1_c1.json 2>/dev/null; cut -c1-230 $O/serving_w1_c1.json
t0=$(date +%s.%N)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.log 2>$O/driver.err
t1=$(date +%s.%N)
grep '^{"metric"' $O/driver.log | tail -1 > $O/bench_driver_command.json
python - $O/bench_driver_command.json $t0 $t1 <<'PY'
import json, sys
raw = open(sys.argv[1]).read()
l = json.loads(raw)
print("driver command: %.1f s wall, line of %d bytes" % (float(sys.argv[3]) - float(sys.argv[2]), len(raw)))
print("value %.4g ms_per_step %.5f" % (l["value"], l["ms_per_step"]))
print("roofline:", json.dumps({k: v for k, v in l["roofline"].items() if not isinstance(v, str)}))
print("config scalars:", json.dumps({k: v for k, v in l["config"].items() if isinstance(v, (int, float)) and not isinstance(v, bool)}))
print("cpu_baseline:", l.get("cpu_baseline", {}).get("value"), l.get("cpu_baseline", {}).get("cores"))
PY
tail -3 $O/driver.err
